#!/bin/bash
# GPU-box job: rocprofv3 kernel trace of the bank-building CLI loop, with and without prefetch -> gpurun_out/bank/bank_build_prof.log
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/bank
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
: > $OUT/bank_build_prof.log
for mode in "" "--no_prefetch"; do
  d=$OUT/trace
  rm -rf $d
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $REPO/tools/bank_build_prof.py run 8 $mode 2>&1 | grep -E "^bank build|rror" | tee -a $OUT/bank_build_prof.log
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python $REPO/tools/bank_build_prof.py read $f | sed "s#$d/##" | tee -a $OUT/bank_build_prof.log
  rm -rf $d
done
