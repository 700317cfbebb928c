"""Calibration probe: what does the vendor GEMM (hipBLASLt via torch.matmul) reach on the ViT's four linear shapes?
Development tool only — the product never calls it.  Run under `rocprofv3 --kernel-trace --stats` to get the Tensile
kernel names (they encode macro-tile, LDS and prefetch configuration)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    t = ops.Timer()
    t.start()
    for _ in range(iters):
        fn()
    t.stop()
    return t.elapsed_ms() / iters


def main():
    M = 64 * 1376
    for (N, K, epi) in [(2048, 1024, 0), (1024, 1024, 2), (4096, 1024, 1), (1024, 4096, 2), (8192, 8192, 0)]:
        m = M if N != 8192 else 8192
        x = torch.randn(m, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device="cuda").to(torch.bfloat16)
        g = torch.ones(N, device="cuda").to(torch.bfloat16)
        r = torch.randn(m, N, device="cuda").to(torch.bfloat16)
        o = torch.empty(m, N, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * m * N * K
        res = []
        for _ in range(3):
            t_lt = timeit(lambda: torch.matmul(x, w.t(), out=o))
            t_me = timeit(lambda: ops.gemm(x, w, b, 0 if N == 8192 else epi, gamma=g, resid=r, out=o))
            res.append((t_lt, t_me))
        lt = min(a for a, _ in res)
        me = min(b_ for _, b_ in res)
        print(f"M={m} N={N} K={K}: hipBLASLt (no epilogue) {lt:.3f} ms = {fl / lt / 1e9:.0f} TF | "
              f"freepose_amd (fused epilogue {epi}) {me:.3f} ms = {fl / me / 1e9:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
