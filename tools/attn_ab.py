"""attention kernel timing, several shapes (in-process medians).  python tools/attn_ab.py [B]"""
import statistics
import sys

import torch

sys.path.insert(0, '.')
from freepose_amd import _lib
import os
if os.environ.get('FP_ATTN_VARIANT') or os.environ.get('FP_LAB_LIB'):
    _lib.use_lab()   # FP_ATTN_VARIANT exists only in the lab build
from freepose_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for n_tok in (1374, 905):
    npad = (n_tok + 15) // 16 * 16
    qk = (torch.randn(B * npad, 2048, device="cuda") * 1.5).to(torch.bfloat16)
    vt = torch.randn(B, 16, 64, npad, device="cuda").to(torch.bfloat16)
    o = torch.empty(B * npad, 1024, device="cuda", dtype=torch.bfloat16)
    res = []
    for rnd in range(7):
        ops.attention(qk, vt, n_tok, out=o)
        torch.cuda.synchronize()
        tm = ops.Timer()
        tm.start()
        for _ in range(10):
            ops.attention(qk, vt, n_tok, out=o)
        tm.stop()
        res.append(tm.elapsed_ms() / 10)
    fl = 4.0 * B * n_tok * n_tok * 1024
    m = statistics.median(res)
    print(f"attention B={B} n={n_tok}: median {m:.3f} ms = {fl / m / 1e9:.0f} TF (best {fl / min(res) / 1e9:.0f})")
