#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or ln_folded" 2>&1 | tail -5
python -m pytest tests/test_gpu_vit.py tests/test_gpu_pose_parity.py tests/test_gpu_lab.py -x -q 2>&1 | tail -8
python tools/lab_selfcheck.py 2>&1 | tail -5
python tools/attn_variant_ab.py 0,64,128 64 2>&1 | tee gpurun_out/r04/attn_pre_ab2.log
python bench.py --steps 2 --warmup 1 --no-config-legs --video-frames 0 > gpurun_out/r04/bench_pre.json 2> gpurun_out/r04/bench_pre.err; tail -c 1500 gpurun_out/r04/bench_pre.json
for cfg in "1 518" "4 518"; do
  set -- $cfg
  rm -rf /tmp/p1
  B=$1 RES=$2 N=20 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o b1 -- python tools/vit_batch_prof.py > gpurun_out/r04/b${1}_${2}_prof.log 2>&1
  find /tmp/p1 -name '*stats*' | head
  f=$(find /tmp/p1 -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/r04/b${1}_${2}_kernel_stats.csv
  grep "per forward" gpurun_out/r04/b${1}_${2}_prof.log
done
