#!/bin/bash
# GPU-box job: rocprofv3 kernel stats of bench.py + PMC passes on the dominant GEMM.  Outputs under gpurun_out/prof.
# (counters are collected in their own passes, never together with sys/hip/hsa tracing)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
V=${FP_GEMM_VARIANT:-}

echo "== kernel trace + stats of bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
find $OUT/bench -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
tail -2 $OUT/bench_stdout.log

for shape in "1024 4096 2" "2048 1024 0" "4096 1024 1"; do
  set -- $shape
  tag="N$1_K$2_e$3"
  echo "== PMC passes for gemm $tag"
  timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $OUT/pmc1_$tag -o p -- python $REPO/tools/gemm_probe.py --N $1 --K $2 --epi $3 --iters 3 > $OUT/pmc1_$tag.log 2>&1
  timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc2_$tag -o p -- python $REPO/tools/gemm_probe.py --N $1 --K $2 --epi $3 --iters 3 > $OUT/pmc2_$tag.log 2>&1
  timeout 300 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE -d $OUT/pmc3_$tag -o p -- python $REPO/tools/gemm_probe.py --N $1 --K $2 --epi $3 --iters 3 > $OUT/pmc3_$tag.log 2>&1
done
# keep only the small CSVs (counter collection + stats), drop bulky traces
python - <<'EOF'
import csv, glob, json, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof"
summary = {}
for d in sorted(glob.glob(out + "/pmc*_N*")):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "gemm_bf16_kernel" not in k:
                continue
            acc["gemm"][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]] += 1
    summary[os.path.basename(d)] = {c: {"sum": v, "dispatch_rows": cnt[c]} for c, v in acc["gemm"].items()}
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
EOF
cd $OUT && find . -name "*.csv" -size +2M -delete; find . -name "*.db" -delete; du -sh .
