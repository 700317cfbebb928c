#!/bin/bash
# GPU-box job: rocprofv3 kernel trace of small-batch ViT-L forwards (B=1 @518, B=5 @420, B=21 @420) -> gpurun_out/small/<tag>.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
TAG=${1:-base}
for cfg in "1 518" "5 420" "21 420" "2 518"; do
  set -- $cfg
  d=$OUT/trace_${TAG}_$1_$2
  rm -rf $d
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $REPO/tools/small_trace.py run $1 $2 10 2>&1 | grep -E "^B=" | tee -a $OUT/${TAG}.txt
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python $REPO/tools/small_trace.py read $f 10 | sed "s#$d/##" | tee -a $OUT/${TAG}.txt
  rm -rf $d
done
