#!/bin/bash
# GPU-box job: rocprofv3 on bench.py.  Pass 1: kernel trace + stats.  Passes 2-4: PMC counters (each in its own run,
# never combined with tracing).  Summaries are written under gpurun_out/prof and copied to profiles/ by hand.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
BENCH="python $REPO/bench.py --no-cpu-baseline --video-frames 0 --no-config-legs --no-cli-legs"
export FP_CSRC_SHA=$(cd $REPO && python -c "import bench; print(bench.csrc_hash())")
echo "csrc_sha16 = $FP_CSRC_SHA"

echo "== pass 1: kernel trace + stats (bench.py --steps 2 --warmup 1)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH --steps 2 --warmup 1 > $OUT/stats_stdout.log 2>&1
find $OUT/stats -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
grep '"metric"' $OUT/stats_stdout.log | tail -1 > $OUT/bench_line_under_rocprof.json

echo "== pass 2: FETCH_SIZE"
timeout 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
echo "== pass 3: WRITE_SIZE"
timeout 900 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
echo "== pass 4: SQ counters + GRBM"
timeout 900 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
echo "== pass 6: VALU / LDS issue activity"
timeout 900 rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES -d $OUT/pmc_valu -o p -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_valu.log 2>&1
echo "== pass 5: L2 hit/miss"
timeout 900 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o p -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_l2.log 2>&1

python - <<'EOF'
import csv, glob, json, os, collections, re
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof"
def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^>]*>)?", k)
    return (m.group(1) + (m.group(2) or "")) if m else k[:60]
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2", "pmc_valu"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            c = per[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
sha = os.environ.get("FP_CSRC_SHA", "")
summary = {k: {c: {"sum": v[0], "dispatches": v[1]} for c, v in cs.items()} for k, cs in per.items()}
summary["_meta"] = {"csrc_sha16": sha, "source": "rocprofv3 --pmc on `bench.py --steps 1 --warmup 0`, one counter group per pass"}
json.dump(summary, open(out + "/pmc_by_kernel.json", "w"), indent=1)
def kernel_summary(prefix, label):
    g = collections.defaultdict(lambda: [0.0, 0])
    for k, cs in per.items():
        if k.startswith(prefix):
            for c, v in cs.items():
                g[c][0] += v[0]; g[c][1] += v[1]
    r = {"kernel": label, "csrc_sha16": sha}
    for c, v in g.items():
        r[c] = v[0]
    if g.get("GRBM_GUI_ACTIVE", [0])[0]:
        simd_cycles = 1024 * g["GRBM_GUI_ACTIVE"][0] / 8          # 1024 SIMDs; GUI_ACTIVE is summed over the 8 XCDs
        if "SQ_VALU_MFMA_BUSY_CYCLES" in g: r["mfma_busy_frac"] = g["SQ_VALU_MFMA_BUSY_CYCLES"][0] / simd_cycles
    if g.get("SQ_WAVE_CYCLES", [0])[0]:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in g: r[c + "_per_wave_cycle"] = g[c][0] / g["SQ_WAVE_CYCLES"][0]
    if g.get("SQ_BUSY_CYCLES", [0])[0]:
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in g: r[c + "_per_sq_busy_cycle"] = g[c][0] / g["SQ_BUSY_CYCLES"][0]
    if "SQ_LDS_BANK_CONFLICT" in g and g.get("SQ_LDS_IDX_ACTIVE", [0])[0]:
        r["lds_conflict_frac"] = g["SQ_LDS_BANK_CONFLICT"][0] / g["SQ_LDS_IDX_ACTIVE"][0]
    return r
json.dump(kernel_summary("attn_fwd_kernel", "attn_fwd_kernel (all attention launches of one bench step)"), open(out + "/attn_pmc.json", "w"), indent=1)
# dominant kernel: all gemm launches (both schedules' kernel names)
g = collections.defaultdict(lambda: [0.0, 0])
for k, cs in per.items():
    if k.startswith("gemm_bf16_kernel") or k.startswith("gemm_asm_kernel"):
        for c, v in cs.items():
            g[c][0] += v[0]; g[c][1] += v[1]
res = {"kernel": "gemm_bf16_kernel (all ViT linear-layer launches of one bench step)", "csrc_sha16": sha, "source": "rocprofv3 --pmc on `bench.py --steps 1 --warmup 0`, separate passes"}
if "FETCH_SIZE" in g and "WRITE_SIZE" in g:
    n = g["FETCH_SIZE"][1]
    fetch = g["FETCH_SIZE"][0] * 1024 * 2     # KiB; gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM section)
    write = g["WRITE_SIZE"][0] * 1024          # KiB; calibrated: equals M*N*2 exactly on the single-GEMM probe
    res.update({"launches": n, "fetch_bytes_per_launch_corrected": fetch / n, "write_bytes_per_launch": write / g["WRITE_SIZE"][1],
                "hbm_bytes_per_launch": fetch / n + write / g["WRITE_SIZE"][1], "fetch_correction": "x2 (gfx950 FETCH_SIZE under-count)"})
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"):
    if c in g:
        res[c] = g[c][0]
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res and res["GRBM_GUI_ACTIVE"]:
    res["mfma_busy_frac"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * res["GRBM_GUI_ACTIVE"] / 8)   # 1024 SIMDs; GUI_ACTIVE summed over 8 XCDs
if "TCC_HIT_sum" in res:
    res["l2_hit_rate"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
json.dump(res, open(out + "/gemm_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
EOF
cd $OUT && find . -name "*.csv" -size +3M -delete; rm -rf stats/*/*trace* 2>/dev/null; du -sh .
head -14 $OUT/bench_kernel_stats.csv | cut -c1-170
