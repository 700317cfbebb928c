#!/bin/bash
# GPU-box job: the vendor library's GEMM kernel (hipBLASLt through torch.matmul: one wave per SIMD, 128x128 per wave, i.e. HALF this
# library's LDS fragment traffic) against gemm_bf16_kernel on the ViT shapes — TFLOP/s (tools/blaslt_calib.py) and PMC: LDS-active
# cycles, MFMA-busy cycles, GRBM_GUI_ACTIVE (-> shader clock = GUI_ACTIVE / 8 / duration).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/vendor
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
python $REPO/tools/blaslt_calib.py 2>&1 | grep "M="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o v -- python $REPO/tools/blaslt_calib.py > $OUT/t.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/p -o v -- python $REPO/tools/blaslt_calib.py > $OUT/p.log 2>&1
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/vendor"
dur = {}
for r in csv.DictReader(open(glob.glob(f"{out}/t/**/*kernel_stats*.csv", recursive=True)[0])):
    dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        c = per[row["Kernel_Name"]][row["Counter_Name"]]
        c[0] += float(row["Counter_Value"]); c[1] += 1
print("kernel | calls | avg us | per dispatch: LDS_IDX_ACTIVE, LDS_BANK_CONFLICT, INSTS_LDS, MFMA_BUSY_CYCLES, GUI_ACTIVE | mfma-busy frac | LDS-active / GUI cycle per CU")
for k, cs in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0])[0]):
    if "gemm" not in k.lower() and "cijk" not in k.lower():
        continue
    g = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    gui = g.get("GRBM_GUI_ACTIVE", 0) / 8
    name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    d = dur.get(k, (0, 0))
    print(f"{name:64s} | {d[0]:4d} | {d[1] / 1e3:8.1f} | {g.get('SQ_LDS_IDX_ACTIVE', 0):.3e} {g.get('SQ_LDS_BANK_CONFLICT', 0):.2e} {g.get('SQ_INSTS_LDS', 0):.3e} "
          f"{g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3e} {gui:.3e} | {g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1024 * gui, 1):.3f} | {g.get('SQ_LDS_IDX_ACTIVE', 0) / max(256 * gui, 1):.3f}"
          + (f" | clock {gui / max(d[1], 1):.2f} GHz" if d[1] else ""))
PY
rm -rf $OUT/t $OUT/p
