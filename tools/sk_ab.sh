#!/bin/bash
# GPU-box job: balanced-tier A/B per layer shape (tools/sk_ab.py) under rocprofv3 --kernel-trace -> gpurun_out/sk/<tag>.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
TAG=${1:-ab}
d=$OUT/trace_$TAG
rm -rf $d
SK_FORMS=${SK_FORMS:-} timeout 1500 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $REPO/tools/sk_ab.py run $OUT/labels_$TAG.json $2 > $OUT/${TAG}_run.log 2>&1
grep -E "check|labels|Error|error|assert" $OUT/${TAG}_run.log | tail -80
f=$(find $d -name "*kernel_trace.csv" | head -1)
python $REPO/tools/sk_ab.py read $f $OUT/labels_$TAG.json | tee $OUT/$TAG.txt
rm -rf $d
