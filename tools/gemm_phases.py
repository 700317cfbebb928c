"""Where a small-tier GEMM launch spends its time, from inside the kernel (lab build, gemm_dbg = 1024: every workgroup's wave 0 stamps the
100 MHz wall clock at entry, when its first K tile has landed, after the K loop and after the epilogue's stores have been issued).
Prints, relative to the earliest entry of the launch: median / max entry delay, first-tile latency, K loop, epilogue, and the span from
the first entry to the last exit.  python tools/gemm_phases.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()
from freepose_amd import ops  # noqa: E402

lib = _lib.load()
g = torch.Generator().manual_seed(1)
ops.set_option("gemm_sk", 1)
for M, N, K, epi, tag in ((4560, 1024, 4096, 2, "B5 fc2"), (4560, 1024, 1024, 2, "B5 proj"), (4560, 2048, 1024, 0, "B5 qk"), (4560, 4096, 1024, 1, "B5 fc1"),
                          (1376, 1024, 4096, 2, "B1 fc2"), (1376, 1024, 1024, 2, "B1 proj"), (1376, 4096, 1024, 1, "B1 fc1")):
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
    ws = [(torch.randn((N, K), generator=g) * 0.03).to(torch.bfloat16).cuda() for _ in range(4)]
    bias = torch.randn((N,), generator=g).to(torch.bfloat16).cuda()
    resid = torch.randn((M, N), generator=g).to(torch.bfloat16).cuda()
    ops.set_option("gemm_variant", 238 | 4096)      # no row split: one kernel per launch
    ops.set_option("gemm_dbg", 1024 | 2048)
    for w in ws:
        ops.gemm(x, w, bias, epi, gamma=bias, resid=resid)
    torch.cuda.synchronize()
    tiles128, tiles64 = -(-M // 128) * -(-N // 128), -(-M // 64) * -(-N // 64)
    nwg = tiles64 if tiles128 < 256 else tiles128
    if -(-M // 256) * -(-N // 256) >= 192:
        print(tag, "runs on the big tier: skipped"); continue
    buf = np.zeros((nwg, 5), dtype=np.uint64)
    _lib.check(lib.fp_lab_read_scratch(ops.context(), buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    t = buf.astype(np.float64) / 100.0          # us
    t0 = t[:, 0].min()
    ent, first, loop, epi_t = t[:, 0] - t0, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"{tag} M={M} N={N} K={K} ({nwg} workgroups, {K // 64} K steps): entry delay med {np.median(ent):.2f} max {ent.max():.2f} | first tile {np.median(first):.2f} (max {first.max():.2f}) | "
          f"shader clock {np.median(buf[:, 4].astype(np.float64) / (t[:, 3] - t[:, 0])):.0f} MHz | K loop {np.median(loop):.2f} (max {loop.max():.2f}) = {np.median(loop) / (K // 64):.3f} us/step | epilogue {np.median(epi_t):.2f} (max {epi_t.max():.2f}) | first entry -> last exit {t[:, 3].max() - t0:.2f} us", flush=True)
    big = np.zeros((65536 + nwg * 8,), dtype=np.uint64)
    _lib.check(lib.fp_lab_read_scratch(ops.context(), big.ctypes.data_as(C.c_void_p), big.nbytes))
    st = big[65536:].reshape(nwg, 8).astype(np.float64)
    d = np.diff(st, axis=1)
    names = ["settle A (earlier frag reads landed)", "issue frag reads B + MFMA block A", "wait: next K tile landed", "barrier", "issue DMA of tile +NS", "settle B", "issue frag reads A' + MFMA block B"]
    print("    K step 8, shader clocks (median over workgroups): " + " | ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names)) + f" | total {np.median(st[:, 7] - st[:, 0]):.0f}", flush=True)
ops.set_option("gemm_dbg", 0)
