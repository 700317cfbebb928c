#!/bin/bash
# GPU-box job: kernel stats of the video tracking step (tools/video_perf.py) + wall time per frame
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/video
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
python $REPO/tools/video_perf.py 2>&1 | grep "video step"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o v -- python $REPO/tools/video_perf.py > $OUT/stdout.log 2>&1
f=$(find $OUT/t -name "*kernel_stats*.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
import os
nobj = [int(x) for x in os.environ.get("VIDEO_OBJECTS", "1").split(",")]
nfr = sum((2 if n == 1 and not os.environ.get("VIDEO_ONLY15") else 1) * (3 + 20) * n for n in nobj)      # frame-objects: (3 warm-up + 20 timed) frames per setting
print(f"sum of kernel time per frame-object: {tot / nfr / 1e6:.2f} ms over {nfr} frame-objects")
for r in rows[:16]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"  {n:60s} calls/frame={int(r['Calls']) / nfr:6.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/frame={float(r['TotalDurationNs']) / nfr / 1e6:7.3f}")
PY
rm -rf $OUT/t
