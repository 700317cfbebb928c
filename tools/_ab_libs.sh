#!/bin/bash
# A/B of two library builds on one box (tools/_lib_old.so vs the tree's build): alternating runs of tools/ab_perf.py
cp freepose_amd/lib/libfreepose_hip.so /tmp/lib_new.so
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then cp tools/_lib_old.so freepose_amd/lib/libfreepose_hip.so; else cp /tmp/lib_new.so freepose_amd/lib/libfreepose_hip.so; fi
    echo "== $which"; timeout 200 python tools/ab_perf.py 238 gemm ${1:-294464} 2>&1 | grep "gemm"
  done
done
cp /tmp/lib_new.so freepose_amd/lib/libfreepose_hip.so
