cd $GRAFT_REPO_ROOT
for vb in 192 288 256 144; do
python bench.py --vit-batch $vb --no-cpu-baseline --video-frames 0 --no-config-legs --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vit-batch', $vb, d['ms_per_step'], d['roofline']['frac'], d['stage_ms_rank0'])"
done
