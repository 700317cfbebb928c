#!/bin/bash
# GPU-box job: rocprofv3 kernel stats of one bench step with the LayerNorm fold off / on (per-kernel-instantiation times)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/lnab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
BENCH="python $REPO/bench.py --no-cpu-baseline --video-frames 0 --steps 1 --warmup 1"
for f in 0 1; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$f -o b -- $BENCH --ln-fused $f > $OUT/stdout$f.log 2>&1
  find $OUT/s$f -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_lnf$f.csv
  rm -rf $OUT/s$f
done
python - <<'PY'
import csv, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/lnab"
for f in (0, 1):
    rows = list(csv.DictReader(open(f"{out}/kernel_stats_lnf{f}.csv")))
    print(f"== ln_fused={f}")
    for r in rows[:22]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        print(f"{n:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
