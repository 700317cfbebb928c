"""What a K step of the small GEMM tiers is made of (lab build, wrong numerics): the plain dispatch on the B = 5 @420^2 and B = 1 @518^2
launch sizes with parts of the software-pipelined loop removed (gemm_dbg: 64 no steady-state DMA, 128 no MFMAs, 256 no fragment
reads, 512 no barrier).  python tools/step_ablate.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()
from freepose_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(1)
ops.set_option("gemm_sk", 1)
for M, N, K, epi, tag in ((4560, 1024, 4096, 2, "B5 fc2"), (4560, 1024, 1024, 2, "B5 proj"), (4560, 2048, 1024, 0, "B5 qk"), (1376, 1024, 4096, 2, "B1 fc2"), (19152, 1024, 1024, 4, "B21 v")):
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
    ws = [(torch.randn((N, K), generator=g) * 0.03).to(torch.bfloat16).cuda() for _ in range(8)]
    bias = torch.randn((N,), generator=g).to(torch.bfloat16).cuda()
    resid = torch.randn((M, N), generator=g).to(torch.bfloat16).cuda()
    row = []
    for name, dbg in (("full", 0), ("no DMA", 64), ("no MFMA", 128), ("no frag reads", 256), ("no barrier", 512), ("no MFMA, no reads", 128 | 256),
                      ("no DMA, no reads", 64 | 256), ("DMA only", 128 | 256 | 512), ("nothing", 64 | 128 | 256 | 512)):
        ops.set_option("gemm_dbg", dbg)
        def call(w):
            if epi == 4:
                return ops.gemm_vt(x, w, bias, 912, 16)
            return ops.gemm(x, w, bias, epi, gamma=bias, resid=resid)
        call(ws[0])
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t = ops.Timer(); t.start()
            for i in range(8):
                call(ws[i])
            t.stop()
            best = min(best, t.elapsed_ms() / 8 * 1e3)
        row.append(f"{name} {best:.1f}")
    ops.set_option("gemm_dbg", 0)
    print(f"{tag} M={M} N={N} K={K}: " + " | ".join(row) + "  (us per call incl. launch gaps)", flush=True)
