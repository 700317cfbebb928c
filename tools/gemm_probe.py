"""Run one GEMM shape repeatedly (for rocprofv3 --pmc passes and quick A/B timing)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=88064)
ap.add_argument("--N", type=int, default=1024)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--epi", type=int, default=2)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
x = torch.randn(a.M, a.K, device="cuda").to(torch.bfloat16)
w = (torch.randn(a.N, a.K, device="cuda") * 0.02).to(torch.bfloat16)
b = torch.zeros(a.N, device="cuda").to(torch.bfloat16)
g = torch.ones(a.N, device="cuda").to(torch.bfloat16)
r = torch.randn(a.M, a.N, device="cuda").to(torch.bfloat16)
o = torch.empty(a.M, a.N, device="cuda", dtype=torch.bfloat16)
ops.gemm(x, w, b, a.epi, gamma=g, resid=r, out=o)
torch.cuda.synchronize()
t = ops.Timer()
t.start()
for _ in range(a.iters):
    ops.gemm(x, w, b, a.epi, gamma=g, resid=r, out=o)
t.stop()
ms = t.elapsed_ms() / a.iters
print(f"gemm M={a.M} N={a.N} K={a.K} epi={a.epi}: {ms:.4f} ms {2.0 * a.M * a.N * a.K / ms / 1e9:.1f} TFLOP/s")
