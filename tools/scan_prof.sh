#!/bin/bash
# GPU-box job: rocprofv3 kernel trace of tools/scan_perf.py; splits bank_scan_kernel dispatches into cold (the dispatch right
# behind an eviction fill) and warm, prints both distributions.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/scan
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
for EV in read write; do
export SCAN_EVICT=$EV
rm -rf $OUT/t
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python $REPO/tools/scan_perf.py > $OUT/stdout.log 2>&1
cat $OUT/stdout.log | grep "Q=\|eviction"
python - <<'PY'
import csv, glob, os, statistics
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/scan"
f = glob.glob(f"{out}/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
cold, warm = {}, {}
prev_fill = False
for r in rows:
    n = r["Kernel_Name"]
    if "bank_scan_kernel" in n:
        key = n.split("bank_scan_kernel")[1].split("(")[0]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        (cold if prev_fill else warm).setdefault(key, []).append(d)
        prev_fill = False
    elif "fill" in n.lower() or "FillFunctor" in n or "reduce" in n.lower() or "Max" in n:
        prev_fill = True
    elif "topk" in n or "select" in n or "merge" in n:
        pass
    else:
        prev_fill = prev_fill
nb = 46037 * 1024 * 2
lines = []
for key in sorted(set(cold) | set(warm)):
    for tag, d in (("cold", cold), ("warm", warm)):
        v = d.get(key, [])
        if v:
            m = statistics.median(v)
            lines.append(f"bank_scan_kernel{key} {tag}: n={len(v)} median {m:.2f} us (min {min(v):.2f}, max {max(v):.2f}) = {nb / m / 1e6:.2f} TB/s")
print("\n".join(lines))
open(out + "/scan_cold_warm_" + os.environ["SCAN_EVICT"] + ".txt", "w").write("\n".join(lines) + "\n")
PY
done
rm -rf $OUT/t
