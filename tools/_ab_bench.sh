#!/bin/bash
# A/B of library builds on one box through the bench step (development probe): alternating runs over tools/_lib_*.so
cp freepose_amd/lib/libfreepose_hip.so /tmp/lib_tree.so
for rep in 1 2 3; do
  for lib in tools/_lib_*.so; do
    cp $lib freepose_amd/lib/libfreepose_hip.so
    echo "== $(basename $lib): $(timeout 300 python bench.py --no-cpu-baseline --video-frames 0 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), "ms/step; gemm frac", round(d["roofline"]["frac"],4))')"
  done
done
cp /tmp/lib_tree.so freepose_amd/lib/libfreepose_hip.so
