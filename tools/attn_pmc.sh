#!/bin/bash
# GPU-box job: PMC counters of the attention kernels (tools/attn_ab.py, both kernels), one counter group per pass
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/attn_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
CMD="python $REPO/tools/attn_ab.py 64"
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
python - <<'PY'
import csv, glob, os, collections, re
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/attn_pmc"
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        k = "attn64" if "attn_fwd64" in k else ("attn32" if "attn_fwd_kernel" in k else None)
        if not k: continue
        c = per[k][row["Counter_Name"]]
        c[0] += float(row["Counter_Value"]); c[1] += 1
for k, cs in per.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} per dispatch {v[0] / max(v[1], 1):.4g}  (dispatches {v[1]})")
PY
