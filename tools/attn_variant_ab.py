"""Attention kernel flavours in one process (lab build): 0 = unscaled q (four waves per SIMD, just-in-time V^T fragments, plain FMAs), 64 = q prescaled by
log2(e)/8 (what the ViT runs), 128 = packed FMAs (shipped until round 4), 32 = the round-3 flavour (three waves per SIMD, whole-tile V^T prefetch).  python tools/attn_variant_ab.py [variants] [B]"""
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
for n_tok in (1374, 905, 1449):     # 1449 = 532^2: the plain (no short tail) kernel
    npad = (n_tok + 15) // 16 * 16
    qk = (torch.randn(B * npad, 2048, device="cuda") * 1.0).to(torch.bfloat16)
    vt = torch.randn(B, 16, 64, npad, device="cuda").to(torch.bfloat16)
    o = torch.empty(B * npad, 1024, device="cuda", dtype=torch.bfloat16)
    # flavours with bit 64 expect q already multiplied by log2(e) / 8 (in the product: folded into the q rows of the qkv weights)
    qk_pre = qk.clone()
    qk_pre[:, :1024] = (qk[:, :1024].float() * (1.4426950408889634 / 8.0)).to(torch.bfloat16)
    cur = {"v": 0}

    def pick(v):
        cur["v"] = max(v, 0)
        ops.set_option("attn_variant", max(v, 0))

    outs = {}
    for v in variants:
        pick(v)
        outs[v] = ops.attention(qk_pre if v & 64 else qk, vt, n_tok, q_prescaled=bool(v & 64)).float().cpu()
    ops.set_option("attn_variant", -1)
    for v in variants[1:]:
        d = outs[v] - outs[variants[0]]
        print(f"n_tok={n_tok}: variant {v} vs {variants[0]}: max |diff| {d.abs().max().item():.3e}, relative L2 {(d.norm() / outs[variants[0]].norm()).item():.3e}")
    ab(f"attention B={B} n={n_tok}", variants, pick,
       lambda: ops.attention(qk_pre if cur["v"] & 64 else qk, vt, n_tok, out=o, q_prescaled=bool(cur["v"] & 64)), 4.0 * B * n_tok * n_tok * 1024, rounds=8)
