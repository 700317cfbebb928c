"""Attention kernel flavours in one process (lab build): 0 = shipped (four waves per SIMD, just-in-time V^T fragments), 32 = the round-3 flavour (three waves
per SIMD, whole-tile V^T prefetch).  python tools/attn_variant_ab.py [variants] [B]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()
from freepose_amd import ops  # noqa: E402
from tools.ab_perf import ab  # noqa: E402

variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,32").split(",")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for n_tok in (1374, 905):
    npad = (n_tok + 15) // 16 * 16
    qk = (torch.randn(B * npad, 2048, device="cuda") * 1.0).to(torch.bfloat16)
    vt = torch.randn(B, 16, 64, npad, device="cuda").to(torch.bfloat16)
    o = torch.empty(B * npad, 1024, device="cuda", dtype=torch.bfloat16)
    outs = {}
    for v in variants:
        ops.set_option("attn_variant", v)
        outs[v] = ops.attention(qk, vt, n_tok).float().cpu()
    ops.set_option("attn_variant", -1)
    for v in variants[1:]:
        print(f"n_tok={n_tok}: variant {v} vs {variants[0]}: max |diff| {(outs[v] - outs[variants[0]]).abs().max().item():.3e}")
    ab(f"attention B={B} n={n_tok}", variants, lambda v: ops.set_option("attn_variant", max(v, 0)),
       lambda: ops.attention(qk, vt, n_tok, out=o), 4.0 * B * n_tok * n_tok * 1024, rounds=8)
