#!/bin/bash
# GPU-box job (lab build): the big-tile GEMM kernels in one process with rotating variant order — 16-wave HIP kernel (bit 8192 = never
# asm), hand-scheduled 4-wave kernel forced (bit 16384), 8-wave geometry forced (bit 65536) — then the asm kernel with its stores /
# its whole epilogue removed (gemm_dbg 16 / 8).   gpurun -- bash tools/gemm_asm_ab.sh
cd ${GRAFT_REPO_ROOT:-.}
python tools/lab_selfcheck.py 2>&1 | tail -1
echo "== 16-wave (8430) | asm 4 waves (16622) | asm 8 waves (65774), full kernels, M = 294 464"
python tools/ab_perf.py 8430,16622,65774 gemm 294464 2>&1 | grep "gemm"
echo "== asm 4-wave kernel forced: gemm_dbg 0 / 16 (no stores) / 8 (no epilogue)"
FP_GEMM_VARIANT=16622 python tools/gemm_dbg_ab.py 0,16,8 294464 2>&1 | grep "gemm N"
