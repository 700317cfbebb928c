"""Kernel-level parity on the MI355X: each HIP kernel vs a plain torch fp32 reference of the same op
(floating-point kernels) through the C ABI (freepose_amd.ops -> libfreepose_hip.so)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 1024, 1024), (1000, 384, 384), (4096, 2048, 1024),
                                   (70000, 1024, 1024), (3000, 4096, 1024), (2600, 1024, 4096), (17, 1152, 384)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_epilogues(M, N, K, epi):
    from freepose_amd import ops
    x, w = _rand((M, K), 1, 1.0), _rand((N, K), 2, 0.05)
    bias, gamma, resid = _rand((N,), 3, 0.5), _rand((N,), 4, 1.0), _rand((M, N), 5, 1.0)
    out = ops.gemm(x, w, bias, epi, gamma=gamma, resid=resid)
    torch.cuda.synchronize()
    # asymmetric operands (random) make a transposed C-write show up as O(1) error
    ref = x.double() @ w.double().t() + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 2:
        ref = resid.double() + gamma.double() * ref
    err = _rel(out, ref)
    assert err < 6e-3, f"gemm epi={epi} M={M} N={N} K={K}: rel err {err}"
    # element-wise: bf16 output rounding + bf16 rounding points of the fused epilogue
    diff = (out.float().cpu() - ref.float()).abs()
    tol = 0.02 * ref.float().abs() + 0.03
    assert (diff <= tol).all(), f"max diff {diff.max().item()}"


@pytest.mark.parametrize("M_rep,N,K", [(1, 64, 64), (1, 256, 64), (4, 256, 64), (4, 256, 256)])   # 128x128 kernel; 256x256 16-wave kernel (>= 192 big tiles; K = 64 is not a shape of the asm loop), 1 and 4 tiles per CU; hand-scheduled 256x256 kernel (K % 128 == 0) with the pipelined epilogue
def test_gelu_table_is_the_direct_formula_on_every_bf16_input(M_rep, N, K):
    """fc1 epilogue: the table GELU against the direct fp32 expression 0.5 x (1 + erf(x / sqrt 2)) (fp_op_gelu: the expression the
    table is filled from, evaluated elementwise) on ALL 65 536 bf16 inputs, in every tile tier — one-hot weights make the
    pre-activation equal the chosen pattern exactly (NaN/Inf and both zeros included); bit-identical.  Against torch's CPU GELU of
    the same bf16 inputs: equal up to the last-place noise of two erf implementations in the cancelling tail (x < -3.5, |gelu| < 1e-3).
    (The alternative main loops of earlier rounds carry the same table; they live in the lab build: tools/lab_selfcheck.py.)"""
    from freepose_amd import ops
    pats = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)          # every bf16 pattern
    x = torch.zeros((65536 * M_rep, K), dtype=torch.bfloat16)
    x[:, 0] = pats.repeat(M_rep)
    w = torch.zeros((N, K), dtype=torch.bfloat16)
    w[:, 0] = 1.0
    bias = torch.zeros((N,), dtype=torch.bfloat16)
    tab = ops.gemm(x, w, bias, 1).cpu()
    pre = ops.gemm(x, w, bias, 0)                        # the bf16 pre-activation the epilogue sees (x * 1 + 0: -0 becomes +0)
    direct = ops.gelu_direct(pre).cpu()
    a, b = tab.view(torch.int16), direct.view(torch.int16)
    finite = ~torch.isnan(pats.float()).repeat(M_rep)
    assert torch.equal(a[finite], b[finite]), "table GELU differs from the direct formula"
    assert torch.isnan(tab.float()[~finite]).all()
    assert (a == a[:, :1]).all()                                                               # every column saw the same input
    col = tab[:65536, 0].float()
    ref = torch.nn.functional.gelu(pats.float()).to(torch.bfloat16).float()
    ok = torch.isfinite(pats.float())
    same = (col[ok] == ref[ok]).float().mean().item()
    assert same > 0.995, same
    assert ((col[ok] - ref[ok]).abs() <= 2e-6 + 2.0 ** -7 * ref[ok].abs()).all()


@pytest.mark.parametrize("B,npad,H", [(1, 272, 6), (3, 912, 16), (2, 1376, 16), (52, 1376, 16)])   # last: persistent 256x256 path
def test_gemm_vt(B, npad, H):
    from freepose_amd import ops
    D = H * 64
    M = B * npad
    x, w, bias = _rand((M, D), 11, 1.0), _rand((D, D), 12, 0.05), _rand((D,), 13, 0.5)
    vt = ops.gemm_vt(x, w, bias, npad, H)
    torch.cuda.synchronize()
    ref = (x.double() @ w.double().t() + bias.double()).reshape(B, npad, H, 64).permute(0, 2, 3, 1)
    assert _rel(vt, ref) < 6e-3


# M = 160: one-wave 64x64 tiles, 3000 / 17: 128x128, 70 000 x 1024 and 52 x 1376: persistent 256x256 (relocated slabs + constants)
@pytest.mark.parametrize("M,N,K", [(160, 1152, 384), (3000, 2048, 1024), (70000, 1024, 1024), (30000, 4096, 1024), (1000, 768, 768)])
@pytest.mark.parametrize("mode", [0, 1])
def test_ln_folded_linear(M, N, K, mode):
    """LayerNorm folded into the consuming GEMM (FP_EPI_LN_BIAS / LN_GELU) vs fp64 torch LayerNorm + Linear (+ GELU) of the same op
    (hub DINOv2 block: attn.qkv(norm1(x)), mlp.fc1(norm2(x))).  Rows get a large common offset and per-row scale so that
    mean >> std for some rows: the colsum correction must cancel the x W' term in fp32, not in bf16."""
    from freepose_amd import ops
    g = torch.Generator().manual_seed(31)
    x = torch.randn((M, K), generator=g) * (0.2 + 3.0 * torch.rand((M, 1), generator=g)) + 4.0 * torch.randn((M, 1), generator=g)
    x[5 % M] = 0.0                                                       # a pad-like all-zero row: var = 0 -> rstd = 1/sqrt(eps), output = b'
    x[:, 7] += 40.0                                                      # a massive-activation channel
    x = x.to(torch.bfloat16)
    w, bias = _rand((N, K), 32, 0.05), _rand((N,), 33, 0.5)
    g_ln, b_ln = (1.0 + 0.3 * torch.randn(K, generator=g)).to(torch.bfloat16), (0.2 * torch.randn(K, generator=g)).to(torch.bfloat16)
    out = ops.ln_linear(x, g_ln, b_ln, w, bias, mode)
    torch.cuda.synchronize()
    y = torch.nn.functional.layer_norm(x.double(), (K,), g_ln.double(), b_ln.double(), 1e-6)
    ref = y @ w.double().t() + bias.double()
    if mode == 1:
        ref = torch.nn.functional.gelu(ref)
    assert _rel(out, ref) < 8e-3, _rel(out, ref)
    diff = (out.float().cpu() - ref.float()).abs()
    tol = 0.02 * ref.float().abs() + 0.04
    assert (diff <= tol).all(), f"max diff {diff.max().item()} at {np.unravel_index(int(diff.argmax()), diff.shape)}"
    # against the SEPARATE LayerNorm kernel + plain GEMM (the reference's rounding points: LN output rounded to bf16 first)
    sep = ops.gemm(ops.layernorm(x, g_ln, b_ln), w, bias, mode)
    assert _rel(out, sep) < 8e-3
    assert _rel(out, ref) <= 1.25 * _rel(sep, ref) + 1e-4, "the folded form must not be less accurate than LN-then-GEMM"


def test_ln_folded_linear_row_scale():
    """rows below n_scaled carry row_scale inside the fold (the ViT's q rows: log2(e) / 8): those outputs are the unscaled ones times
    the factor up to one bf16 rounding, the others keep their bits"""
    from freepose_amd import ops
    M, N, K, n_scaled = 3000, 2048, 1024, 1024
    g = torch.Generator().manual_seed(37)
    x = (torch.randn((M, K), generator=g) * (0.2 + 3.0 * torch.rand((M, 1), generator=g)) + 2.0 * torch.randn((M, 1), generator=g)).to(torch.bfloat16)
    w, bias = _rand((N, K), 38, 0.05), _rand((N,), 39, 0.5)
    g_ln, b_ln = (1.0 + 0.3 * torch.randn(K, generator=g)).to(torch.bfloat16), (0.2 * torch.randn(K, generator=g)).to(torch.bfloat16)
    plain = ops.ln_linear(x, g_ln, b_ln, w, bias, 0)
    scaled = ops.ln_linear(x, g_ln, b_ln, w, bias, 0, n_scaled=n_scaled, row_scale=ops.ATTN_QSCALE)
    torch.cuda.synchronize()
    assert torch.equal(plain[:, n_scaled:], scaled[:, n_scaled:])
    y = torch.nn.functional.layer_norm(x.double(), (K,), g_ln.double(), b_ln.double(), 1e-6)
    ref = (y @ w.double().t() + bias.double())[:, :n_scaled] * ops.ATTN_QSCALE
    assert _rel(scaled[:, :n_scaled], ref) < 8e-3
    assert _rel(scaled[:, :n_scaled], ref) <= 1.25 * _rel(plain[:, :n_scaled].double().cpu() * ops.ATTN_QSCALE, ref) + 1e-3


@pytest.mark.parametrize("B,npad,H", [(1, 272, 6), (3, 912, 16), (52, 1376, 16)])
def test_ln_folded_vt(B, npad, H):
    from freepose_amd import ops
    D, M = H * 64, B * npad
    g = torch.Generator().manual_seed(41)
    x = (torch.randn((M, D), generator=g) * (0.2 + 2.0 * torch.rand((M, 1), generator=g)) + 2.0 * torch.randn((M, 1), generator=g)).to(torch.bfloat16)
    w, bias = _rand((D, D), 42, 0.05), _rand((D,), 43, 0.5)
    g_ln, b_ln = (1.0 + 0.3 * torch.randn(D, generator=g)).to(torch.bfloat16), (0.2 * torch.randn(D, generator=g)).to(torch.bfloat16)
    vt = ops.ln_linear(x, g_ln, b_ln, w, bias, 2, npad=npad, heads=H)
    torch.cuda.synchronize()
    y = torch.nn.functional.layer_norm(x.double(), (D,), g_ln.double(), b_ln.double(), 1e-6)
    ref = (y @ w.double().t() + bias.double()).reshape(B, npad, H, 64).permute(0, 2, 3, 1)
    assert _rel(vt, ref) < 8e-3, _rel(vt, ref)


@pytest.mark.parametrize("M,N,K", [(160, 384, 384), (3000, 1024, 1024), (70000, 1024, 1024), (2600, 1024, 4096), (900, 768, 3072)])
def test_gemm_emits_row_statistics(M, N, K):
    """producer side of the folded LayerNorm: the LayerScale + residual epilogue also writes per-64-column (sum, sum of squares)
    of its bf16 OUTPUT rows; finalised (mean, rstd) vs torch on those rows, and the output itself equal to the plain epilogue's bits"""
    from freepose_amd import ops
    x, w = _rand((M, K), 51, 1.0), _rand((N, K), 52, 0.05)
    bias, gamma, resid = _rand((N,), 53, 0.5), _rand((N,), 54, 1.0), _rand((M, N), 55, 2.0)
    resid = (resid.float() + 3.0 * torch.randn((M, 1), generator=torch.Generator().manual_seed(56))).to(torch.bfloat16)
    out, stat = ops.gemm_stats(x, w, bias, gamma, resid)
    plain = ops.gemm(x, w, bias, 2, gamma=gamma, resid=resid)
    torch.cuda.synchronize()
    assert torch.equal(out, plain), "the statistics must not change the stored rows"
    o = out.double().cpu()
    mean, var = o.mean(dim=1), o.var(dim=1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-6)
    st = stat.double().cpu()
    # mean and sigma come back as the two-piece bf16 splits the consuming GEMM's init MFMA reads: 16 mantissa bits
    sigma = torch.sqrt(var + 1e-6)
    assert ((st[:, 0] - mean).abs() <= 3e-5 * (1e-3 + mean.abs())).all(), (st[:, 0] - mean).abs().max()
    assert ((st[:, 1] - sigma).abs() <= 3e-5 * sigma).all(), ((st[:, 1] - sigma).abs() / sigma).max()
    assert ((st[:, 2] - rstd).abs() <= 2e-4 * rstd).all(), ((st[:, 2] - rstd).abs() / rstd).max()
    # tile-tier independence: partials are per 64-column block whatever the tile, so a sub-batch reproduces the same bits
    m2 = min(M, 160)
    out2, stat2 = ops.gemm_stats(x[:m2], w, bias, gamma, resid[:m2])
    assert torch.equal(out2, out[:m2]) and torch.equal(stat2, stat[:m2])


@pytest.mark.parametrize("M", [19152, 21 * 912 + 37, 16384 + 128, 6 * 1376, 5 * 912, 8192 + 64])   # the last three: 128x128 rounds + a one-wave-tile remainder
def test_row_split_dispatch_is_invisible(M):
    """launch sizes between the tile tiers (the video path's ~20-crop batches) run whole rounds of the resident grid on 256x256
    tiles and the remaining rows on the finer tiers (gemm_bf16.hip launch_epi).  Rows are independent and every tier produces the
    same bits, so the outputs — incl. the row statistics and the LayerNorm-folded epilogues — must equal those of the unsplit
    dispatch (fp_ctx_set_option "gemm_row_split" = 0) exactly."""
    from freepose_amd import ops
    K = 1024
    x = _rand((M, K), 61, 1.0)
    res = {}
    for tag, split in (("split", -1), ("whole", 0)):
        ops.set_option("gemm_row_split", split)
        try:
            out = {}
            for N in (1024, 2048):
                w, bias = _rand((N, K), 62 + N, 0.05), _rand((N,), 63, 0.5)
                gamma, resid = _rand((N,), 64, 1.0), _rand((M, N), 65, 2.0)
                g_ln, b_ln = _rand((K,), 66, 1.0), _rand((K,), 67, 0.3)
                out[f"bias{N}"] = ops.gemm(x, w, bias, 0)
                out[f"lsres{N}"] = ops.gemm(x, w, bias, 2, gamma=gamma, resid=resid)
                o, st = ops.gemm_stats(x, w, bias, gamma, resid)
                out[f"stats{N}"], out[f"stat_rows{N}"] = o, st
                out[f"ln{N}"] = ops.ln_linear(x, g_ln, b_ln, w, bias, mode=0)
                out[f"ln_gelu{N}"] = ops.ln_linear(x, g_ln, b_ln, w, bias, mode=1)
            torch.cuda.synchronize()
            res[tag] = out
        finally:
            ops.set_option("gemm_row_split", -1)
    for k in res["split"]:
        assert torch.equal(res["split"][k], res["whole"][k]), k
    assert torch.isfinite(res["split"]["stat_rows1024"]).all()


# (ViT-L fc2, a ragged-column shape, ViT-B fc2)   LN-folded epilogues at long K: tools/lab_selfcheck.py (forced tier)
@pytest.mark.parametrize("M,N,K", [(49152 + 37, 1024, 4096), (50000, 1152, 2048), (66000, 768, 3072)])
def test_hand_scheduled_tier_gives_the_bits_of_the_small_tiers(M, N, K):
    """Long-K launches that fill the chip run the hand-scheduled one-wave-per-SIMD kernel (gemm_asm.hip: accumulators in the AGPR file,
    pipelined epilogue, descriptor-clipped ragged rows / columns).  Same K order, same init MFMA, same rounding points as every other
    tier, so its outputs — plain, GELU, LayerScale + residual, row statistics, LayerNorm-folded — must equal, bit for bit, the same
    rows computed in 4096-row slices (which take the 128x128 kernel); plus a float64 reference check of the plain product."""
    from freepose_amd import ops
    x = _rand((M, K), 71, 1.0)
    w, bias = _rand((N, K), 72, 0.03), _rand((N,), 73, 0.5)
    gamma, resid = _rand((N,), 74, 1.0), _rand((M, N), 75, 2.0)
    g_ln, b_ln = _rand((K,), 76, 1.0), _rand((K,), 77, 0.3)

    def run(xs, rs):
        o = {"bias": ops.gemm(xs, w, bias, 0), "gelu": ops.gemm(xs, w, bias, 1), "lsres": ops.gemm(xs, w, bias, 2, gamma=gamma, resid=rs)}
        if K <= 1536:      # the LayerNorm statistics kernel normalises rows of up to 1536 features (every DINOv2 width)
            o["ln"], o["ln_gelu"] = ops.ln_linear(xs, g_ln, b_ln, w, bias, mode=0), ops.ln_linear(xs, g_ln, b_ln, w, bias, mode=1)
        o["stats"], o["stat_rows"] = ops.gemm_stats(xs, w, bias, gamma, rs)
        return o
    whole = run(x, resid)
    parts = [run(x[i:i + 4096], resid[i:i + 4096]) for i in range(0, M, 4096)]
    torch.cuda.synchronize()
    for k in whole:
        assert torch.equal(whole[k], torch.cat([p_[k] for p_ in parts])), k
    rows = torch.arange(0, M, 997)
    ref = x[rows].double() @ w.double().t() + bias.double()
    assert _rel(whole["bias"][rows.cuda()], ref) < 6e-3


# 70 / 129 / 261 / 905 / 1374: the last K/V tile's valid keys fit its first half -> the "short tail first" kernel (attention.hip);
# 97 (33 keys in the tail), 64 and 17 take the plain kernel; 2 tiles (70, 97, 129 -> 3) exercise the shortest loops of both;
# 1449 = 532^2 (41 keys in the last of 23 tiles): the plain kernel at ViT-L size — its pre-scaled instantiation spilled 10 registers
# until round 5 (attention.hip: the first tile is peeled at compile time; tests/test_capi_cpu.py asserts 0 scratch on all four)
@pytest.mark.parametrize("B,H,n_tok", [(1, 6, 261), (2, 16, 905), (2, 16, 1374), (1, 16, 64), (1, 2, 17), (1, 2, 70), (1, 2, 97),
                                       (2, 3, 129), (1, 16, 1449)])
@pytest.mark.parametrize("prescaled", [False, True])
def test_attention(B, H, n_tok, prescaled):
    """prescaled: the q columns hold q log2(e) / 8 in bf16 (what the ViT's folded qkv layer writes) and the kernel takes base-2
    exponents straight from the S^T accumulators; the reference is the softmax of exactly those bf16 values"""
    from freepose_amd import ops
    npad = (n_tok + 15) // 16 * 16
    D = H * 64
    qkv = _rand((B, npad, 3, H, 64), 21, 1.5)
    if prescaled:
        qkv[:, :, 0] = (qkv[:, :, 0].float() * ops.ATTN_QSCALE).to(torch.bfloat16)
    qk = qkv[:, :, :2].reshape(B * npad, 2 * D).contiguous()
    vt = qkv[:, :, 2].permute(0, 2, 3, 1).contiguous()  # [B,H,64,npad]
    o = ops.attention(qk, vt, n_tok, q_prescaled=prescaled)
    torch.cuda.synchronize()
    q, k, v = (qkv[:, :n_tok, i].permute(0, 2, 1, 3).double() for i in range(3))  # [B,H,n,64]
    ref = torch.softmax(q @ k.transpose(-1, -2) * (float(np.log(2.0)) if prescaled else 1.0 / 8.0), dim=-1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(B, n_tok, D)
    got = o.reshape(B, npad, D)[:, :n_tok].float().cpu()
    assert torch.isfinite(o.float()).all(), "pad rows must stay finite"
    err = _rel(got, ref)
    assert err < 1e-2, f"attention rel err {err}"
    assert (got - ref.float()).abs().max().item() < 0.05


@pytest.mark.parametrize("prescaled", [False, True])
def test_attention_forced_rescale(prescaled):
    """spike one key against one query at a late tile so the running max jumps (online-softmax rescale path); a second query whose
    logits all sit far BELOW zero (the prescaled kernel's initial reference) must not underflow to 0 / 0"""
    from freepose_amd import ops
    B, H, n_tok = 1, 1, 300
    npad = 304
    qkv = _rand((B, npad, 3, H, 64), 31, 0.3).float()
    qkv[0, 5, 0, 0] = 4.0          # query 5
    qkv[0, 250, 1, 0] = 4.0        # key 250 (4th tile): q.k = 1024 -> /8 = 128
    qkv[0, :, 1, 0, 1] = 6.0       # every key: component 1 = 6
    qkv[0, 9, 0, 0, 1] = -48.0     # query 9: logits ~ -288 / 8 = -36 ... (x 64 dims of noise); prescaled: ~ -52 in log2 units
    qkv[0, 11, 0, 0, 1] = -400.0   # query 11: logits ~ -300, far below exp2's underflow when taken against a reference of 0
    qkv = qkv.to(torch.bfloat16)
    if prescaled:
        qkv[:, :, 0] = (qkv[:, :, 0].float() * ops.ATTN_QSCALE).to(torch.bfloat16)
    qk = qkv[:, :, :2].reshape(B * npad, 128).contiguous()
    vt = qkv[:, :, 2].permute(0, 2, 3, 1).contiguous()
    o = ops.attention(qk, vt, n_tok, q_prescaled=prescaled).float().cpu().reshape(npad, 64)
    assert torch.isfinite(o).all()
    q, k, v = (qkv[0, :n_tok, i, 0].double() for i in range(3))
    ref = torch.softmax(q @ k.t() * (float(np.log(2.0)) if prescaled else 1.0 / 8.0), dim=-1) @ v
    assert (o[:n_tok] - ref.float()).abs().max().item() < 0.02
    assert (o[5] - v[250].float()).abs().max().item() < 0.02  # query 5 attends (almost) only to key 250


@pytest.mark.parametrize("B,H,W,ps,kp", [(2, 224, 224, 14, 640), (3, 420, 420, 14, 640), (2, 518, 518, 14, 640), (1, 28, 70, 14, 592),
                                         (2, 32, 48, 16, 768), (1, 35, 21, 7, 152)])
def test_im2col_norm_bit_exact(B, H, W, ps, kp):
    """normalise + patch unfold vs the same two bf16 operations in torch (torchvision Normalize on a bf16 tensor: dino.py:12,16) and an
    unfold: every element equal, for the DINOv2 patch size and two others."""
    from freepose_amd import ops
    g = torch.Generator().manual_seed(77)
    img = torch.rand((B, 3, H, W), generator=g).to(torch.bfloat16).cuda()
    got = ops.im2col_norm(img, ps, kp)
    mean = torch.tensor([0.485, 0.456, 0.406]).to(torch.bfloat16).cuda().view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).to(torch.bfloat16).cuda().view(1, 3, 1, 1)
    x = (img - mean) / std                                           # two bf16 ops, each rounded
    K = 3 * ps * ps
    p = x.unfold(2, ps, ps).unfold(3, ps, ps).permute(0, 2, 3, 1, 4, 5).reshape(B * (H // ps) * (W // ps), K)
    ref = torch.zeros((p.shape[0], kp), dtype=torch.bfloat16, device="cuda")
    ref[:, :K] = p
    assert torch.equal(got, ref)


@pytest.mark.parametrize("rows,D", [(5, 384), (1000, 1024), (33, 768)])
def test_layernorm(rows, D):
    from freepose_amd import ops
    x, g, b = _rand((rows, D), 41, 2.0), _rand((D,), 42, 1.0), _rand((D,), 43, 0.5)
    y = ops.layernorm(x, g, b, 1e-6)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), g.float(), b.float(), 1e-6)
    assert (y.float().cpu() - ref).abs().max().item() < 0.03 + 0.01 * ref.abs().max().item()
    assert _rel(y, ref) < 5e-3
