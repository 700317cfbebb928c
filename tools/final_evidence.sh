#!/bin/bash
# GPU-box job: everything the round's evidence needs from ONE box — GPU suite, smoke, the default bench line, then the rocprofv3
# passes of tools/profile_job.sh (kernel stats + PMC) on the same tree.   gpurun --timeout 3600 -- bash tools/final_evidence.sh
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
(timeout 1700 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v amdgpu.ids | tail -40) > gpurun_out/r06/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke.log 2>&1; tail -1 gpurun_out/r06/smoke.log
python bench.py > gpurun_out/r06/bench_line.json 2> gpurun_out/r06/bench_line.err
tail -3 gpurun_out/r06/final_tests.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["stage_ms_rank0"])
PY
bash tools/profile_job.sh 2>&1 | tail -40
# randomised parity at 150 cases per section on the same tree (tests/test_gpu_fuzz.py; the default suite above runs 6 per section)
if [ "${FUZZ:-1}" = "1" ]; then
  (FP_FUZZ_ITERS=${FUZZ_ITERS:-150} timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | grep -v amdgpu.ids | tail -3) > gpurun_out/r06/fuzz_final.log
  tail -1 gpurun_out/r06/fuzz_final.log
fi
# BASELINE config 4 at its own size (576 hypotheses, 518^2), oracle ViT in the reference's bf16 regime and in fp32: ~12 min of host CPU
if [ "${FULL_PARITY:-0}" = "1" ]; then
  (FP_PARITY_QUERIES=6 FP_PARITY_FP32=1 timeout 2400 python -u -m pytest tests/test_gpu_zz_pose_parity_full.py -x -q -s 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06/pose_parity_full.log
  tail -22 gpurun_out/r06/pose_parity_full.log
fi
