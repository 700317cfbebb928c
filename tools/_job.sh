cd $GRAFT_REPO_ROOT
timeout 1400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/scan_perf.py 2>&1 | grep "Q="
python bench.py --no-cpu-baseline --video-frames 0 --steps 2 --warmup 1 > gpurun_out/r04/bench_legs.json 2> gpurun_out/r04/bench_legs.err; tail -c 3500 gpurun_out/r04/bench_legs.json; tail -5 gpurun_out/r04/bench_legs.err
