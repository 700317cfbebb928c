cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vit.py tests/test_gpu_fullsize.py tests/test_gpu_lab.py tests/test_gpu_pose_parity.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -4
python bench.py --no-cpu-baseline --video-frames 0 --no-config-legs --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['stage_ms_rank0'])"
