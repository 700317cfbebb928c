"""Within-process interleaved A/B of kernel variants (boxes differ by +-10 %, so variants are compared in ONE process,
round-robin, reporting median and min).  Development probe."""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()   # measurement variants / hooks live in libfreepose_hip_lab.so only (python -m freepose_amd.build --lab)
from freepose_amd import ops  # noqa: E402


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    t = ops.Timer()
    t.start()
    for _ in range(iters):
        fn()
    t.stop()
    return t.elapsed_ms() / iters


def ab(name, variants, setter, fn, flops, rounds=6):
    res = {v: [] for v in variants}
    for rd in range(rounds):
        # the order rotates every round: the variant that runs first after a switch measured ~3 % slow (clock / cache state left by
        # its predecessor), which biased fixed-order comparisons against the first-listed variant
        order = variants[rd % len(variants):] + variants[:rd % len(variants)]
        for v in order:
            setter(v)
            timeit(fn, iters=2)            # settle
            res[v].append(timeit(fn))
    setter(-1)
    out = []
    for v in variants:
        med, mn = statistics.median(res[v]), min(res[v])
        out.append(f"{v}: {med:.3f} ms ({flops / med / 1e9:.0f} TF, best {flops / mn / 1e9:.0f})")
    print(f"{name:42s} " + " | ".join(out), flush=True)


def main():
    gemm_variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "6,14,0".split(","))]
    only_gemm = len(sys.argv) > 2
    attn_variants = [2, 3, 4]
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 64 * 1376
    for (N, K, epi) in [(2048, 1024, 0), (1024, 1024, 2), (4096, 1024, 1), (1024, 4096, 2)]:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device="cuda").to(torch.bfloat16)
        g = torch.ones(N, device="cuda").to(torch.bfloat16)
        r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ab(f"gemm N={N} K={K} epi={epi}", gemm_variants, lambda v: ops.set_option("gemm_variant", v),
           lambda: ops.gemm(x, w, b, epi, gamma=g, resid=r, out=o), 2.0 * M * N * K)
    if True:   # V projection with the transposed per-head store
        x = torch.randn(M, 1024, device="cuda").to(torch.bfloat16)
        w = (torch.randn(1024, 1024, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.zeros(1024, device="cuda").to(torch.bfloat16)
        vt = torch.empty(M // 1376 if M % 1376 == 0 else M // 912, 16, 64, 1376 if M % 1376 == 0 else 912, device="cuda", dtype=torch.bfloat16)
        ab("gemm_vt N=1024 K=1024", gemm_variants, lambda v: ops.set_option("gemm_variant", v),
           lambda: ops.gemm_vt(x, w, b, 1376 if M % 1376 == 0 else 912, 16, out=vt), 2.0 * M * 1024 * 1024)
    for (B, n_tok) in ([] if only_gemm else [(64, 1374), (64, 905)]):
        npad = (n_tok + 15) // 16 * 16
        qk = torch.randn(B * npad, 2048, device="cuda").to(torch.bfloat16)
        vt = torch.randn(B, 16, 64, npad, device="cuda").to(torch.bfloat16)
        o = torch.empty(B * npad, 1024, device="cuda", dtype=torch.bfloat16)
        ab(f"attention B={B} n={n_tok}", attn_variants, lambda v: ops.set_option("attn_slots", v),
           lambda: ops.attention(qk, vt, n_tok, out=o), 4.0 * B * n_tok * n_tok * 1024)


if __name__ == "__main__":
    main()
