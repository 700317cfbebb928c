"""In-process A/B of FP_GEMM_DBG measurement bits (option "gemm_dbg") on the ViT GEMM shapes.  python tools/gemm_dbg_ab.py 0,512 [M]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()   # measurement variants / hooks live in libfreepose_hip_lab.so only (python -m freepose_amd.build --lab)
from freepose_amd import ops  # noqa: E402
from tools.ab_perf import ab  # noqa: E402

variants = [int(x) for x in sys.argv[1].split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 294464
shapes = [(2048, 1024, 0), (1024, 1024, 2), (4096, 1024, 1), (1024, 4096, 2)]
if len(sys.argv) > 3:
    shapes = [shapes[int(i)] for i in sys.argv[3].split(",")]
for (N, K, epi) in shapes:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device="cuda").to(torch.bfloat16)
    g = torch.ones(N, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ab(f"gemm N={N} K={K} epi={epi}", variants, lambda v: ops.set_option("gemm_dbg", max(v, 0)),
       lambda: ops.gemm(x, w, b, epi, gamma=g, resid=r, out=o), 2.0 * M * N * K, rounds=7)
