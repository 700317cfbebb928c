cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python tools/small_batch_ab.py 518 2>&1 | grep "^B="
python tools/small_batch_ab.py 420 2>&1 | grep "^B="
S=$(date +%s); python bench.py > gpurun_out/r04/bench_line.json 2> gpurun_out/r04/bench_line.err; E=$(date +%s); echo "bench wall: $((E-S)) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["stage_ms_rank0"], d["n_gpus"], d["backend"], d["library"])
print({k: (round(v["value"], 1), v.get("roofline", {}).get("frac")) for k, v in d["configs"].items()})
print(d["video_workload"]["ms_per_frame_per_gpu"], d["video_workload"]["multi_object"]["ms_per_frame_object_per_gpu"])
print(d["cpu_baseline"] and {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")})
PY
