"""Self-check of the LAB build (libfreepose_hip_lab.so, `python -m freepose_amd.build --lab`): the measurement variants the A/B
tools switch between still compute what the product computes.  Run by tests/test_gpu_lab.py in its own process (the product and the
lab library are never loaded together).  python tools/lab_selfcheck.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()
from freepose_amd import ops  # noqa: E402


def main():
    assert _lib.is_lab() and hasattr(_lib.load(), "fp_lab_set_option")
    g = torch.Generator().manual_seed(5)
    # 1. table GELU == direct formula through every GEMM main-loop variant (0: plain loops + direct erff, 6: pipelined 8-wave /
    #    128x128 kernels with the table, default 238: the product's kernels), all 65 536 bf16 inputs, 128x128 and 256x256 tiers
    pats = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    for M_rep, N in ((1, 64), (4, 256)):
        x = torch.zeros((65536 * M_rep, 64), dtype=torch.bfloat16)
        x[:, 0] = pats.repeat(M_rep)
        w = torch.zeros((N, 64), dtype=torch.bfloat16)
        w[:, 0] = 1.0
        bias = torch.zeros((N,), dtype=torch.bfloat16)
        outs = {}
        for var in (-1, 6, 0, 14, 110):
            ops.set_option("gemm_variant", var)
            outs[var] = ops.gemm(x, w, bias, 1).cpu().view(torch.int16)
        ops.set_option("gemm_variant", -1)
        finite = ~torch.isnan(pats.float()).repeat(M_rep)
        for var, o in outs.items():
            assert torch.equal(o[finite], outs[-1][finite]), f"gemm_variant {var} differs (N={N})"
    # 2. every variant of the plain / LayerScale+residual GEMM gives the product's bits on a ragged shape
    M, N, K = 50000, 1024, 1024          # fills the chip with 256x256 tiles, ragged last row tile
    x = (torch.randn((M, K), generator=g)).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16)
    bias, gamma = torch.randn((N,), generator=g).to(torch.bfloat16), torch.randn((N,), generator=g).to(torch.bfloat16)
    resid = torch.randn((M, N), generator=g).to(torch.bfloat16)
    ref0, ref2 = ops.gemm(x, w, bias, 0).cpu(), ops.gemm(x, w, bias, 2, gamma=gamma, resid=resid).cpu()
    for var in (0, 6, 14, 46, 110, 238 | 256, 238 | 2048, 238 | 512, 238 | 8192, 238 | 16384, 238 | 65536):
        ops.set_option("gemm_variant", var)
        assert torch.equal(ops.gemm(x, w, bias, 0).cpu(), ref0), var
        assert torch.equal(ops.gemm(x, w, bias, 2, gamma=gamma, resid=resid).cpu(), ref2), var
    ops.set_option("gemm_variant", -1)
    # 2b. the hand-scheduled kernel forced on for every shape it supports (bit 16384) == never (bit 8192): LayerNorm-folded epilogues,
    #     table GELU and row statistics through the pipelined epilogue at K = 1024, where the product keeps the 16-wave kernel
    g_ln, b_ln = torch.randn((K,), generator=g).to(torch.bfloat16), (torch.randn((K,), generator=g) * 0.3).to(torch.bfloat16)
    outs = {}
    for var in (238 | 8192, 238 | 16384, 238 | 65536):
        ops.set_option("gemm_variant", var)
        st_o, st_r = ops.gemm_stats(x, w, bias, gamma, resid)
        outs[var] = [ops.gemm(x, w, bias, 1).cpu(), ops.ln_linear(x, g_ln, b_ln, w, bias, mode=0).cpu(),
                     ops.ln_linear(x, g_ln, b_ln, w, bias, mode=1).cpu(), st_o.cpu(), st_r.cpu()]
    ops.set_option("gemm_variant", -1)
    for a_, b_ in list(zip(outs[238 | 8192], outs[238 | 16384])) + list(zip(outs[238 | 8192], outs[238 | 65536])):
        assert torch.equal(a_, b_), "hand-scheduled tier differs from the 16-wave kernel"
    # 2c. the 64x64 tier's K-tile ring: every depth cap (2 = the double buffer; 8 = deeper than the product ever picks) gives the same bits
    #     — plain, LayerNorm-folded + GELU, row statistics; K = 1024 and the long K = 4096 walk
    for (Ms, Ns, Ks) in ((1376, 1024, 1024), (300, 1024, 4096), (160, 2048, 384)):
        xs = torch.randn((Ms, Ks), generator=g).to(torch.bfloat16)
        ws = (torch.randn((Ns, Ks), generator=g) * 0.05).to(torch.bfloat16)
        bs, gs = torch.randn((Ns,), generator=g).to(torch.bfloat16), torch.randn((Ns,), generator=g).to(torch.bfloat16)
        rs = torch.randn((Ms, Ns), generator=g).to(torch.bfloat16)
        gl, bl = torch.randn((Ks,), generator=g).to(torch.bfloat16), (torch.randn((Ks,), generator=g) * 0.3).to(torch.bfloat16)
        outs = {}
        for cap in (2, 3, 4, 6, 8):
            ops.set_option("gemm_ring", cap)
            so, sr = ops.gemm_stats(xs, ws, bs, gs, rs)
            outs[cap] = [ops.gemm(xs, ws, bs, 0).cpu(), so.cpu(), sr.cpu()]
            if Ks <= 1536:
                outs[cap].append(ops.ln_linear(xs, gl, bl, ws, bs, mode=1).cpu())
        ops.set_option("gemm_ring", -1)
        for cap in (3, 4, 6, 8):
            for a_, b_ in zip(outs[2], outs[cap]):
                assert torch.equal(a_, b_), f"K-tile ring cap {cap} differs from the double buffer at {(Ms, Ns, Ks)}"
    # 2d. round 5's small-tier forms (profiles/r05_ab.md): the 128x128 tier on 8 waves, the spread DMA issue and the hand-scheduled
    #     128x128 kernel walk K in the product's order -> the product's bits; the balanced (stream-K) forms add fp32 partial tiles in K
    #     order -> run-to-run identical, within the fp32 summation-order distance of the product (<= 1 bf16 ulp of each rounding point)
    for (Ms, Ns, Ks) in ((4560, 1024, 4096), (1376, 2048, 1024), (2768, 1024, 1024)):
        xs = torch.randn((Ms, Ks), generator=g).to(torch.bfloat16)
        ws = (torch.randn((Ns, Ks), generator=g) * 0.05).to(torch.bfloat16)
        bs, gs = torch.randn((Ns,), generator=g).to(torch.bfloat16), torch.randn((Ns,), generator=g).to(torch.bfloat16)
        rs = torch.randn((Ms, Ns), generator=g).to(torch.bfloat16)
        gl, bl = torch.randn((Ks,), generator=g).to(torch.bfloat16), (torch.randn((Ks,), generator=g) * 0.3).to(torch.bfloat16)

        def run():
            so, sr = ops.gemm_stats(xs, ws, bs, gs, rs)
            r = [ops.gemm(xs, ws, bs, 0).cpu(), ops.gemm(xs, ws, bs, 2, gamma=gs, resid=rs).cpu(), so.cpu(), sr.cpu()]
            if Ks <= 1536:
                r += [ops.ln_linear(xs, gl, bl, ws, bs, mode=0).cpu(), ops.ln_linear(xs, gl, bl, ws, bs, mode=1).cpu()]
            return r
        ref = run()
        for var in (238 | 262144, 238 | 524288, 238 | 524288 | 262144, 238 | 1048576, 238 | 1048576 | 4096):
            ops.set_option("gemm_variant", var)
            for a_, b_ in zip(ref, run()):
                assert torch.equal(a_, b_), f"gemm_variant {var} differs from the product at {(Ms, Ns, Ks)}"
        ops.set_option("gemm_variant", -1)
        for mode, grid in ((2, 256), (2, 512), (3, 1024), (4, 256)):
            ops.set_option("gemm_sk", mode)
            ops.set_option("gemm_sk_grid", grid)
            o1, o2 = run(), run()
            for i, (a_, b_, c_) in enumerate(zip(ref, o1, o2)):
                assert torch.equal(b_, c_), f"balanced tier {mode}/{grid} is not run-to-run deterministic at {(Ms, Ns, Ks)}"
                if i != 3:      # (row statistics: compared through the outputs they are computed from)
                    tol = (a_.float().abs() + (rs.float().abs() if i in (1, 2) else 0.0)) * 2.0 ** -6 + 1e-3
                    assert ((b_.float() - a_.float()).abs() <= tol).all(), f"balanced tier {mode}/{grid} off at {(Ms, Ns, Ks)} output {i}"
        ops.set_option("gemm_sk", -1)
        ops.set_option("gemm_sk_grid", -1)
    # 3. attention: ring depths and the short-tail-off flavour agree to rounding (the exponent reference differs by tile order only)
    B, H, n_tok = 3, 16, 905
    npad = (n_tok + 15) // 16 * 16
    qk = (torch.randn((B * npad, 2 * H * 64), generator=g) * 0.5).to(torch.bfloat16)
    vt = torch.randn((B, H, 64, npad), generator=g).to(torch.bfloat16)
    base = ops.attention(qk, vt, n_tok).float().cpu()
    ops.set_option("attn_variant", 128)          # packed FMAs (the form shipped until round 4): the same bits as the plain ones
    assert torch.equal(ops.attention(qk, vt, n_tok).float().cpu(), base)
    ops.set_option("attn_variant", -1)
    for name, val in (("attn_slots", 3), ("attn_slots", 4), ("attn_variant", 8), ("attn_variant", 32)):
        ops.set_option(name, val)
        o = ops.attention(qk, vt, n_tok).float().cpu()
        ops.set_option(name, -1)
        assert (o - base).abs().max().item() <= 2e-2, (name, val)
    # 4. measurement hooks are reachable here (and only here): loop-only GEMM runs without writing its output
    out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    big = x.repeat(20, 1)
    outb = torch.full((big.shape[0], N), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.set_option("gemm_dbg", 8)
    ops.gemm(big, w, bias, 0, out=outb)
    ops.set_option("gemm_dbg", -1)
    torch.cuda.synchronize()
    assert (outb == 7.0).all(), "gemm_dbg = 8 (no epilogue) still wrote the output"
    print("LAB_SELFCHECK_OK")


if __name__ == "__main__":
    main()
