#!/bin/bash
# GPU-box job: per-kernel durations of the rasteriser launches (rocprofv3 --kernel-trace --stats) for the bench's mesh
#   tools/raster_kernels_prof.sh [sub=6] [tiled=1]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/raster_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o rp -- python $REPO/tools/raster_prof.py ${1:-6} ${2:-1} > $OUT/stdout.log 2>&1
f=$(find $OUT -name "*kernel_stats*.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.3f} ms  {r['Percentage']} %")
PY
