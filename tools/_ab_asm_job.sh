cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
echo "== asm (238) vs 16-wave HIP kernel (8430), full kernels"; python tools/ab_perf.py 238,8430 gemm 294464 2>&1 | grep "gemm N"
echo "== asm kernel: dbg 0 / 16 (no stores) / 8 (no epilogue)"; python tools/gemm_dbg_ab.py 0,16,8 294464 2>&1 | grep "gemm N"
