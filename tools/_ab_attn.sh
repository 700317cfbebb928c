#!/bin/bash
# A/B of library builds on one box through tools/attn_ab.py (development probe): alternating runs over tools/_lib_*.so
cp freepose_amd/lib/libfreepose_hip.so /tmp/lib_tree.so
for rep in 1 2 3; do
  for lib in tools/_lib_*.so; do
    cp $lib freepose_amd/lib/libfreepose_hip.so
    echo "== $(basename $lib)"; timeout 200 python tools/attn_ab.py ${1:-64} 2>&1 | grep attention
  done
done
cp /tmp/lib_tree.so freepose_amd/lib/libfreepose_hip.so
