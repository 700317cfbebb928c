cd $GRAFT_REPO_ROOT
timeout 600 python tools/lab_selfcheck.py 2>&1 | tail -2
python tools/ab_perf.py 8430,16622,65774 gemm 294464 2>&1 | grep "gemm"
