#!/bin/bash
# GPU-box job: per-instantiation GEMM times of one bench step for FP_GEMM_DBG settings (experiment, wrong numerics for dbg != 0)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/gdbg
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
BENCH="python $REPO/bench.py --lab --no-cpu-baseline --video-frames 0 --steps 1 --warmup 1"   # FP_GEMM_DBG exists in the lab build only
for d in ${DBGS:-0 1 2 4 7}; do
  FP_GEMM_DBG=$d timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$d -o b -- $BENCH > $OUT/stdout$d.log 2>&1
  f=$(find $OUT/s$d -name "*kernel_stats*.csv" | head -1)
  echo "== FP_GEMM_DBG=$d"
  python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    print(f"  {n:58s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
  rm -rf $OUT/s$d
done
