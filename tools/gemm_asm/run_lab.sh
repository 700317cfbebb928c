#!/bin/bash
# GPU-box job (stage 1): the hand-scheduled one-wave-per-SIMD K loop (tools/gemm_asm) against the library's main
# loop without its epilogue (FP_GEMM_DBG=8 / lab build) on the bench's GEMM shapes, random operands, with socket power and shader
# clock polled while each shape loops, and SQ_LDS_IDX_ACTIVE per flop from a separate PMC pass.
#   gpurun -- bash tools/gemm_asm/run_lab.sh        (variants: RATES="1.0,2.0 1.0,1.0 ...")
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/gemm_asm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LOG=$OUT/r04_gemm_asm_loop.log
: > $LOG

poll_start() {  # background sampler of socket power / sclk
  ( while true; do rocm-smi -P -c --json 2>/dev/null | tr -d '\n'; echo; sleep 0.05; done ) > $OUT/smi_$1.jsonl &
  POLL=$!
}
poll_stop() {
  kill $POLL 2>/dev/null; wait $POLL 2>/dev/null
  python3 - "$OUT/smi_$1.jsonl" <<'PY'
import json, statistics, sys
pw, ck = [], []
for ln in open(sys.argv[1]):
    try:
        d = json.loads(ln); card = d[sorted(d)[0]]
    except Exception:
        continue
    for k, v in card.items():
        kl = k.lower()
        if "power" in kl and "(w)" in kl:
            try: pw.append(float(v))
            except Exception: pass
        if kl.startswith("sclk") and "mhz" in str(v).lower():
            ck.append(float(str(v).lower().replace("(", "").replace(")", "").replace("mhz", "").strip()))
# the upper half of the samples = while the kernel loops (the binary also runs checks / fills)
pw.sort(); ck.sort()
hp = pw[len(pw) // 2:] or [float("nan")]
print(f"    power median-of-upper-half {statistics.median(hp):.0f} W (max {max(hp):.0f}), sclk median {statistics.median(ck) if ck else float('nan'):.0f} MHz, {len(pw)} samples")
PY
}

M=$((214 * 1376))
SHAPES="$M 2048 1024 $M 1024 1024 $M 4096 1024 $M 1024 4096"
for rates in ${RATES:-"1.0,2.0"}; do
  r0=${rates%,*}; r1=${rates#*,}
  python3 $REPO/tools/gemm_asm/gen_loop.py --rate0 $r0 --rate1 $r1 > $REPO/tools/gemm_asm/loop_body.inc
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $REPO/tools/gemm_asm/gemm_asm_lab.hip -o $OUT/lab_$rates || { echo "build failed ($rates)" | tee -a $LOG; continue; }
  echo "== asm loop, filler rates phase0/phase1 = $rates (random operands, loop only, 3 s per shape)" | tee -a $LOG
  for sh in "$M 2048 1024" "$M 1024 1024" "$M 4096 1024" "$M 1024 4096"; do
    poll_start x
    timeout 300 $OUT/lab_$rates --secs 3 $sh 2>&1 | grep -v "^check" | tee -a $LOG
    poll_stop x | tee -a $LOG
  done
done
FIRST=$(echo ${RATES:-"1.0,2.0"} | awk '{print $1}')
echo "== asm loop, zero operands ($FIRST)" | tee -a $LOG
timeout 300 $OUT/lab_$FIRST --zero --secs 1 $SHAPES 2>&1 | grep -v "^check" | tee -a $LOG

echo "== library main loop without its epilogue (FP_GEMM_DBG=8), same shapes (tools/gemm_power.py)" | tee -a $LOG
cd $REPO && FP_GEMM_DBG=8 PYTHONPATH=$REPO timeout 600 python3 tools/gemm_power.py 2>&1 | tail -12 | tee -a $LOG
cd /tmp

echo "== PMC: LDS-active cycles, MFMA-busy cycles, wave stalls (one launch set per kernel; separate pass)" | tee -a $LOG
timeout 600 rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/p -o v -- $OUT/lab_$FIRST $SHAPES > $OUT/p.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o v -- $OUT/lab_$FIRST $SHAPES > $OUT/t.log 2>&1
python3 - <<PY | tee -a $LOG
import csv, glob, collections
out = "$OUT"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_asm" in row["Kernel_Name"]:
            per[row["Dispatch_Id"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
# dispatches in order: 3 checks, then per shape 3 warm-up + 10 timed
M = $M
shapes = [(M, 2048, 1024), (M, 1024, 1024), (M, 4096, 1024), (M, 1024, 4096)]
ids = sorted(per, key=int)
body = ids[3:]
for si, (m, n, k) in enumerate(shapes):
    d = body[si * 13 + 5] if len(body) > si * 13 + 5 else None
    if d is None: break
    g = {c: sum(v) for c, v in per[d].items()}
    fl = 2.0 * m * n * k
    gui = g.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"  asm N={n} K={k}: LDS_IDX_ACTIVE/flop {g.get('SQ_LDS_IDX_ACTIVE', 0) / fl:.3e}  bank-conflict cycles {g.get('SQ_LDS_BANK_CONFLICT', 0):.2e}  "
          f"MFMA-busy {g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1024 * gui, 1):.3f}  WAIT_ANY/WAVE_CYCLES {g.get('SQ_WAIT_ANY', 0) / max(g.get('SQ_WAVE_CYCLES', 1), 1):.3f}  "
          f"WAIT_INST_ANY/WAVE_CYCLES {g.get('SQ_WAIT_INST_ANY', 0) / max(g.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
rm -rf $OUT/p $OUT/t $OUT/lab_* $OUT/smi_*
echo done
