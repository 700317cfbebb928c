#!/bin/bash
# GPU-box job: the round-5 small-tier measurements whose output only went to the console while they were developed -> gpurun_out/r05/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
export PYTHONPATH=$REPO
python tools/step_ablate.py 2>&1 | grep -E "^B" > $OUT/step_ablate.log
python tools/gemm_phases.py 2>&1 | grep -E "^B|K step" > $OUT/gemm_phases.log
(python tools/small_policy_ab.py 420 1,2,3,5,8,12,21; python tools/small_policy_ab.py 518 1,2,3,6,8) 2>&1 | grep -E "^B=" > $OUT/small_policy_ab.log
hipcc --offload-arch=gfx950 -O3 tools/experiments/fill_stride_probe.hip -o /tmp/fsp 2>/dev/null && (/tmp/fsp; /tmp/fsp private) > $OUT/fill_stride_probe.log 2>&1
tail -n +1 $OUT/*.log | cut -c1-400
