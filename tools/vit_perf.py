"""Quick on-GPU timing of the ViT forward and its kernel classes (development probe, not the bench contract)."""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dinov2_vitl14_reg")
    ap.add_argument("--res", type=int, default=518)
    ap.add_argument("--batches", default="1,16,64")
    ap.add_argument("--layer", type=int, default=22)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    vit = ops.ViT(args.model, seed=0)
    out = {}
    for B in [int(b) for b in args.batches.split(",")]:
        img = torch.rand(B, 3, args.res, args.res, device="cuda").to(torch.bfloat16)
        vit(img, layer=args.layer, feature_type="patch")
        torch.cuda.synchronize()
        t = ops.Timer()
        t.start()
        for _ in range(args.iters):
            vit(img, layer=args.layer, feature_type="patch")
        t.stop()
        ms = t.elapsed_ms() / args.iters
        fl = vit.flops(B, args.res, args.res, args.layer)
        vit.profile(True)
        vit(img, layer=args.layer, feature_type="patch")
        pr = vit.profile_read()
        vit.profile(False)
        res = {"ms": ms, "crops_per_s": B / ms * 1e3, "tflops": fl / ms / 1e9,
               "gemm_ms": pr["ms_gemm"], "gemm_tflops": pr["gemm_flops"] / max(pr["ms_gemm"], 1e-9) / 1e9,
               "attn_ms": pr["ms_attn"], "other_ms": pr["ms_other"]}
        out[B] = res
        print(f"B={B}: {json.dumps(res)}", flush=True)
    # GEMM microbench at the ViT-L shapes (M = 64 crops x 1376 rows)
    M = 64 * 1376
    for (N, K, epi) in [(2048, 1024, 0), (1024, 1024, 2), (4096, 1024, 1), (1024, 4096, 2)]:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device="cuda").to(torch.bfloat16)
        g = torch.ones(N, device="cuda").to(torch.bfloat16)
        r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(x, w, b, epi, gamma=g, resid=r, out=o)
        torch.cuda.synchronize()
        t = ops.Timer()
        t.start()
        for _ in range(5):
            ops.gemm(x, w, b, epi, gamma=g, resid=r, out=o)
        t.stop()
        ms = t.elapsed_ms() / 5
        print(f"gemm M={M} N={N} K={K} epi={epi}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)
    # attention microbench
    for (B, n_tok) in [(64, 1374), (64, 905)]:
        npad = (n_tok + 15) // 16 * 16
        qk = torch.randn(B * npad, 2048, device="cuda").to(torch.bfloat16)
        vt = torch.randn(B, 16, 64, npad, device="cuda").to(torch.bfloat16)
        o = torch.empty(B * npad, 1024, device="cuda", dtype=torch.bfloat16)
        ops.attention(qk, vt, n_tok, out=o)
        torch.cuda.synchronize()
        t = ops.Timer()
        t.start()
        for _ in range(5):
            ops.attention(qk, vt, n_tok, out=o)
        t.stop()
        ms = t.elapsed_ms() / 5
        print(f"attention B={B} n={n_tok}: {ms:.3f} ms  {4.0 * B * n_tok * n_tok * 1024 / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
