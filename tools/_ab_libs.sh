#!/bin/bash
# A/B of several library builds on one box: alternating runs of tools/gemm_dbg_ab.py over tools/_lib_*.so (development probe)
#   bash tools/_ab_libs.sh "<shape indices>" [M]
cp freepose_amd/lib/libfreepose_hip.so /tmp/lib_tree.so
for rep in 1 2 3; do
  for lib in tools/_lib_*.so; do
    cp $lib freepose_amd/lib/libfreepose_hip.so
    echo "== $(basename $lib)"; timeout 200 python tools/gemm_dbg_ab.py 0 ${2:-294464} ${1:-0,1,2,3} 2>&1 | grep "gemm"
  done
done
cp /tmp/lib_tree.so freepose_amd/lib/libfreepose_hip.so
