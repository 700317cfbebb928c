"""HIP retrieval / pose kernels vs the C oracle (oracle/fp_oracle.c) — bit-exact where the work is integer, byte or
canonically-ordered fp32 ("dot64"), through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bank(N, D, seed, shared=2.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    mu = rng.standard_normal(D).astype(np.float32)
    x = rng.standard_normal((N, D)).astype(np.float32) + shared * mu
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("N,D", [(1000, 1024), (100, 384), (46037, 1024), (777, 768)])
def test_bank_prepare_and_topk_bit_exact(N, D):
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    bank = _bank(N, D, 21)
    qs = _bank(5, D, 22)
    bank_o = fo.bank_prepare(bank)
    bank_g = ops.bank_prepare(torch.from_numpy(bank))
    assert np.array_equal(fo.torch_to_bits(bank_g), bank_o), "bank prep (cast + bf16 normalise) differs"
    q_bits = fo.l2norm_rows(fo.to_bf16_bits(qs))
    k = min(100, N)
    s_o, i_o = fo.bank_topk(bank_o, q_bits, k)
    s_g, i_g = ops.bank_topk(bank_g, fo.bits_to_torch(q_bits), k)
    assert np.array_equal(i_g.cpu().numpy(), i_o), "retrieved indices must be bit-exact (score desc, index asc)"
    assert np.array_equal(s_g.cpu().numpy().view(np.uint32), s_o.view(np.uint32)), "scores must be bit-exact"
    # the tie rule is exercised: with a shared mean the 100th place is always tied in bf16
    if N >= 10000:
        full = fo.bank_scores(bank_o, q_bits[0])
        assert (full == s_o[0, -1]).sum() > 1


def test_row_norms_next_to_a_bf16_rounding_boundary():
    """F.normalize on bf16 rows rounds the norm to bf16 before dividing: a square root that is one fp32 ulp off moves the norm across a
    rounding boundary for about one row in 16 000, and then EVERY element of the row.  1.5 M short rows hold ~100 such rows; all of
    them must carry the oracle's bits (the device code needs an IEEE square root, not `__fsqrt_rn` = v_sqrt_f32: csrc/common.h)."""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    g = torch.Generator().manual_seed(77)
    rows, D = 1_500_000, 64
    xt = (torch.randn((rows, D), generator=g) * (0.5 + 7.5 * torch.rand((rows, 1), generator=g))).to(torch.bfloat16)
    x = fo.torch_to_bits(xt)
    want = fo.l2norm_rows(x)
    got = fo.torch_to_bits(ops.l2_normalize(xt))
    bad = np.unique(np.argwhere(got != want)[:, 0])
    assert len(bad) == 0, f"{len(bad)} rows differ, first {bad[:5]}"
    # how many rows were at risk: norm within 2 fp32 ulps of a bf16 midpoint
    n = np.sqrt((fo.bits_to_torch(x).double().numpy() ** 2).sum(1))
    frac = np.abs((n.astype(np.float32).view(np.uint32) & 0xFFFF).astype(np.int64) - 0x8000)
    assert (frac <= 2).sum() > 20


def test_topk_tie_heavy_and_offsets():
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    N, D = 5000, 1024
    rng = np.random.Generator(np.random.PCG64(5))
    base = _bank(50, D, 6)
    bank = base[rng.integers(0, 50, size=N)]          # only 50 distinct rows -> massive ties
    bank_o = fo.bank_prepare(bank)
    q = fo.l2norm_rows(fo.to_bf16_bits(_bank(3, D, 7)))
    for k in (1, 3, 100, 1024):
        s_o, i_o = fo.bank_topk(bank_o, q, k, idx_offset=12345)
        s_g, i_g = ops.bank_topk(fo.bits_to_torch(bank_o), fo.bits_to_torch(q), k, idx_offset=12345)
        assert np.array_equal(i_g.cpu().numpy(), i_o)
        assert np.array_equal(s_g.cpu().numpy(), s_o)


def test_topk_degenerate_full_bank_of_equal_keys():
    """46 037 identical rows (and an all-zero query: every score 0, and a NaN query): every key equals the threshold, so the
    "== T" count of the ordered compaction reaches 46 037 >= 2^15 — the packed (count>, count==) scan must not go negative
    (round-2 advisor finding).  Canonical order then is index ascending."""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    N, D = 46037, 1024
    row = _bank(1, D, 61)
    bank_o = fo.bank_prepare(np.repeat(row, N, axis=0))
    bank_g = fo.bits_to_torch(bank_o).cuda()
    qs = fo.l2norm_rows(fo.to_bf16_bits(_bank(2, D, 62)))
    zero = np.zeros((1, D), dtype=np.uint16)
    nan = np.full((1, D), 0x7fc0, dtype=np.uint16)
    q = np.concatenate([qs, zero, nan], axis=0)
    for k in (1, 100, 1024):
        s_o, i_o = fo.bank_topk(bank_o, q, k)
        s_g, i_g = ops.bank_topk(bank_g, fo.bits_to_torch(q), k)
        assert np.array_equal(i_g.cpu().numpy(), i_o)
        assert np.array_equal(s_g.cpu().numpy().view(np.uint32), s_o.view(np.uint32))
        assert np.array_equal(i_o[0], np.arange(k))


def test_topk_merge_matches_unsharded():
    """bank-row sharding: per-shard top-k + merge == global top-k"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    N, D, k, R = 4000, 1024, 100, 4
    bank_o = fo.bank_prepare(_bank(N, D, 31))
    q = fo.l2norm_rows(fo.to_bf16_bits(_bank(6, D, 32)))
    bank_g, q_g = fo.bits_to_torch(bank_o).cuda(), fo.bits_to_torch(q).cuda()
    s_ref, i_ref = ops.bank_topk(bank_g, q_g, k)
    cs, ci = [], []
    for r in range(R):
        lo, hi = r * N // R, (r + 1) * N // R
        s, i = ops.bank_topk(bank_g[lo:hi].contiguous(), q_g, k, idx_offset=lo)
        cs.append(s)
        ci.append(i)
    s_m, i_m = ops.topk_merge(torch.cat(cs, 1), torch.cat(ci, 1), k)
    assert torch.equal(i_m, i_ref) and torch.equal(s_m, s_ref)
    s_o, i_o = fo.topk_merge(torch.cat(cs, 1).cpu().numpy(), torch.cat(ci, 1).cpu().numpy(), k)
    assert np.array_equal(i_m.cpu().numpy(), i_o)


@pytest.mark.parametrize("B,gh,gw,D,cell", [(3, 30, 30, 1024, 14), (2, 16, 16, 384, 14), (4, 1, 900, 1024, 1)])
def test_ffa_bit_exact(B, gh, gw, D, cell):
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    rng = np.random.Generator(np.random.PCG64(41))
    P = gh * gw
    feats = fo.to_bf16_bits(rng.standard_normal((B, P, D)).astype(np.float32))
    if cell == 1:
        mask = (rng.random((B, P)) < 0.4).astype(np.uint8)
        mask_o = mask.reshape(B, 1, P)
    else:
        mask = np.zeros((B, gh * cell, gw * cell), np.uint8)
        yy, xx = np.mgrid[0:gh * cell, 0:gw * cell]
        for b in range(B):
            cy, cx, ry, rx = rng.uniform(0.3, 0.7) * gh * cell, rng.uniform(0.3, 0.7) * gw * cell, rng.uniform(0.1, 0.4) * gh * cell, rng.uniform(0.1, 0.4) * gw * cell
            mask[b] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.uint8)
        mask_o = mask
    ob, of = fo.ffa(feats, mask_o, cell)
    g = ops.ffa(fo.bits_to_torch(feats), torch.from_numpy(mask), cell=cell)
    assert np.array_equal(fo.torch_to_bits(g), ob)
    g32 = ops.ffa(fo.bits_to_torch(feats), torch.from_numpy(mask), cell=cell, out_f32=True)
    assert np.array_equal(g32.cpu().numpy(), of)
    gn = ops.ffa(fo.bits_to_torch(feats), torch.from_numpy(mask), cell=cell, normalize=True)
    assert np.array_equal(fo.torch_to_bits(gn), fo.l2norm_rows(ob))


def test_ffa_fused_normalisation_equals_the_row_kernel():
    """up to 16 crops the masked-mean kernel normalises the rows itself (the last column-slab workgroup of a crop to arrive, one
    agent-scope counter per crop); above, the row kernel does.  Same bits for the same crops — at mask storage of every byte alignment,
    sparse single-pixel masks (one byte at a cell's corner), over repeated calls (the arrival counters return to zero), beside the oracle."""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    rng = np.random.Generator(np.random.PCG64(43))
    gh = gw = 37
    P, D, cell = gh * gw, 1024, 14
    nbig = 20
    feats = fo.to_bf16_bits(rng.standard_normal((nbig, P, D)).astype(np.float32))
    mask = (rng.random((nbig, gh * cell, gw * cell)) < 0.0004).astype(np.uint8)          # a few isolated pixels per crop
    mask[1] = 0
    mask[1, 13, 13] = 1                                                                   # last byte of the first cell
    mask[2] = 0
    mask[2, -1, -1] = 1                                                                   # the very last byte of a crop's mask
    mask[3, 100:300, 50:400] = 1
    ob, of = fo.ffa(feats, mask, cell)
    ft, mt = fo.bits_to_torch(feats).cuda(), torch.from_numpy(mask).cuda()
    big = ops.ffa(ft, mt, cell=cell, normalize=True)                                      # 20 crops: cell mask, masked mean, row normalisation
    big_raw = ops.ffa(ft, mt, cell=cell)
    assert np.array_equal(fo.torch_to_bits(big_raw.cpu()), ob) and np.array_equal(fo.torch_to_bits(big.cpu()), fo.l2norm_rows(ob))
    flat = torch.zeros(mt.numel() + 8, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        for lo, n in ((0, 1), (1, 3), (4, 16), (19, 1)):                                  # 1 ... 16 crops: normalised by the masked-mean kernel
            for off in ((0, 1, 2, 3) if n == 1 else (rep,)):                              # mask storage at every byte alignment
                m = flat[off:off + n * mask[0].size].view(n, gh * cell, gw * cell)
                m.copy_(mt[lo:lo + n])
                got = ops.ffa(ft[lo:lo + n], m, cell=cell, normalize=True)
                assert torch.equal(got.view(torch.int16), big[lo:lo + n].view(torch.int16)), (rep, lo, n, off)
                raw = ops.ffa(ft[lo:lo + n], m, cell=cell)
                assert torch.equal(raw.view(torch.int16), big_raw[lo:lo + n].view(torch.int16))
                f32 = ops.ffa(ft[lo:lo + n], m, cell=cell, out_f32=True)
                assert np.array_equal(f32.cpu().numpy()[~np.isnan(of[lo:lo + n])], of[lo:lo + n][~np.isnan(of[lo:lo + n])])


def test_ffa_empty_mask_is_nan():
    """0/0 -> NaN like feat[mask].mean(0) on an empty selection (extract_retrieval_features.py:59)"""
    from freepose_amd import ops
    f = torch.randn(1, 16, 384).to(torch.bfloat16)
    out = ops.ffa(f, torch.zeros(1, 16, dtype=torch.uint8), cell=1, out_f32=True)
    assert torch.isnan(out).all()


@pytest.mark.parametrize("T,P,D", [(7, 900, 1024), (19, 256, 384), (600, 900, 1024)])
def test_template_score_bit_exact(T, P, D):
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    rng = np.random.Generator(np.random.PCG64(51))
    if T == 600:
        T_o = 24  # oracle is scalar C: check a slice bit-exactly, the rest for range
    else:
        T_o = T
    tm = fo.to_bf16_bits((rng.standard_normal((T, P, D)) * 3).astype(np.float32))
    q = fo.l2norm_rows(fo.to_bf16_bits(rng.standard_normal((P, D)).astype(np.float32)))
    s_g = ops.template_score(fo.bits_to_torch(tm), fo.bits_to_torch(q)).cpu().numpy()
    s_o = fo.template_score(tm[:T_o], q)
    assert np.array_equal(s_g[:T_o].view(np.uint32), s_o.view(np.uint32))
    assert np.isfinite(s_g).all() and np.abs(s_g).max() <= 1.0
    # known answer: a template equal to the query scores ~1 and wins
    tm2 = tm.copy()
    tm2[3] = q
    s2 = ops.template_score(fo.bits_to_torch(tm2), fo.bits_to_torch(q)).cpu().numpy()
    assert s2.argmax() == 3 and s2[3] > 0.99
    # mask-weighted variant (online_pose_estimator.py:69-74)
    w = rng.random((T_o, P)).astype(np.float32)
    sw_g = ops.template_score(fo.bits_to_torch(tm[:T_o]), fo.bits_to_torch(q), torch.from_numpy(w)).cpu().numpy()
    sw_o = fo.template_score(tm[:T_o], q, w)
    assert np.array_equal(sw_g.view(np.uint32), sw_o.view(np.uint32))
    # pre-normalised store: normalise the rows ONCE (in place), then the streaming-dot scorer — the same bits
    # as the on-the-fly scorer on ALL T templates, and as the oracle (which normalises per call like pose_estimator.py:85)
    tn = ops.l2_normalize(fo.bits_to_torch(tm).cuda().clone(), inplace=True)
    assert np.array_equal(fo.torch_to_bits(tn[:T_o].cpu()).reshape(-1, D), fo.l2norm_rows(tm[:T_o].reshape(-1, D)))
    s_n = ops.template_score(tn, fo.bits_to_torch(q), normalized=True).cpu().numpy()
    assert np.array_equal(s_n.view(np.uint32), s_g.view(np.uint32))
    sw_n = ops.template_score(tn[:T_o], fo.bits_to_torch(q), torch.from_numpy(w), normalized=True).cpu().numpy()
    assert np.array_equal(sw_n.view(np.uint32), sw_o.view(np.uint32))


def test_crop_resize_pad_bit_exact():
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    rng = np.random.Generator(np.random.PCG64(61))
    H, W = 480, 640
    img = rng.random((1, 3, H, W)).astype(np.float32)
    boxes = []
    for _ in range(40):
        x0, y0 = rng.integers(0, W - 20), rng.integers(0, H - 20)
        boxes.append([x0, y0, rng.integers(x0 + 6, W + 1), rng.integers(y0 + 6, H + 1)])
    boxes += [[0, 0, W, H], [10, 10, 210, 210], [5, 7, 305, 207], [100, 50, 521, 471], [0, 0, 6, 6]]
    boxes = np.array(boxes, dtype=np.int32)
    masks = (rng.random((len(boxes), H, W)) < 0.7).astype(np.uint8)
    for ext in (0.0, 0.05, 0.1, 0.2):
        for mode in (0, 1, 2):
            o = fo.crop_resize_pad(img, boxes, 420, ext, masks, mode)
            g = ops.crop_resize_pad(torch.from_numpy(img), torch.from_numpy(boxes), 420, ext, torch.from_numpy(masks), mode)
            assert np.array_equal(g.cpu().numpy(), o), f"crop ext={ext} mode={mode}"
    # u8 HWC source, per-box images (render -> crop path) and bf16 output
    imgs = rng.integers(0, 256, size=(4, 420, 420, 3), dtype=np.uint8)
    bx = np.array([[100, 120, 300, 333], [0, 0, 420, 420], [150, 10, 260, 400], [105, 105, 314, 314]], np.int32)
    o = fo.crop_resize_pad(imgs, bx, 420, 0.0)
    g = ops.crop_resize_pad(torch.from_numpy(imgs), torch.from_numpy(bx), 420, 0.0)
    assert np.array_equal(g.cpu().numpy(), o)
    gb = ops.crop_resize_pad(torch.from_numpy(imgs), torch.from_numpy(bx), 420, 0.0, out_bf16=True)
    assert torch.equal(gb.cpu(), torch.from_numpy(o).to(torch.bfloat16))


def test_geodesic_and_rotations():
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    R = ops.generate_rotations(20000)
    Ro = fo.generate_rotations(20000)
    assert np.allclose(R, Ro, atol=1e-15)
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-12)
    grid = torch.from_numpy(R).cuda()
    sizes = []
    for i in (0, 17, 5000, 19999):
        idx_g = ops.geodesic_select(grid, R[i], 15.0)
        idx_o = fo.geodesic_select(R, R[i], 15.0)
        assert np.array_equal(idx_g, idx_o) and i in idx_g
        sizes.append(len(idx_g))
    assert 10 <= min(sizes) and max(sizes) <= 30  # 16-23 neighbours


def _icosphere(sub=3):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(sub):
        cache, nf = {}, []
        vl = list(v)

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (vl[a] + vl[b]) / 2
                vl.append(m / np.linalg.norm(m))
                cache[key] = len(vl) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(vl), np.array(nf)
    return v, f


@pytest.mark.parametrize("sub,W", [(2, 420), (4, 420), (3, 200)])
def test_rasterizer_bit_exact(sub, W):
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    v, f = _icosphere(sub)
    rng = np.random.Generator(np.random.PCG64(71))
    v = v * (1 + 0.25 * np.sin(3 * v[:, :1]) * np.cos(2 * v[:, 1:2]))  # displaced icosphere
    v /= np.abs(v).max()
    colors = rng.integers(0, 256, size=(len(v), 3), dtype=np.uint8)
    Rs = fo.generate_rotations(8)
    poses = np.tile(np.eye(4, dtype=np.float32), (8, 1, 1))
    poses[:, :3, :3] = Rs
    poses[:, :3, 3] = [0, 0, 1.1]
    poses[7, :3, 3] = [0.3, -0.2, 0.9]   # partly off-screen
    fx = 600.0 * W / 420
    rgb_o, d_o = fo.rasterize(v, f, colors, poses, 0.25, fx, fx, W / 2, W / 2, W, W)
    mesh = ops.Mesh(v, f, colors)
    rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(poses), 0.25, fx, fx, W / 2, W / 2, W, W)
    assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32)), "depth must be bit-exact"
    assert np.array_equal(rgb_g.cpu().numpy(), rgb_o), "rgb must be bit-exact"
    cov = (d_o > 0).mean(axis=(1, 2))
    assert (cov[:7] > 0.1).all()
    # known answer: silhouette of a sphere-like object at z=1.1 spans about 2*0.25*600/1.1 px
    ext_o = fo.depth_extents(d_o, fx, fx, W / 2, W / 2)
    ext_g = ops.depth_extents(d_g, fx, fx, W / 2, W / 2).cpu().numpy()
    assert np.array_equal(ext_g, ext_o)
    wpx = ext_o[0, 2] - ext_o[0, 0]
    assert abs(wpx - 2 * 0.25 * fx / 1.1) < 0.25 * 2 * 0.25 * fx / 1.1


def test_rasterizer_large_triangles_and_untextured():
    """two big triangles (wave-per-triangle queue path) + white default colour"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], [-0.5, -0.5, -0.3], [0.5, -0.5, -0.3], [0, 0.7, -0.3]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3], [4, 6, 5]], np.int32)
    poses = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    poses[:, 2, 3] = 1.1
    poses[1, :3, :3] = fo.generate_rotations(5)[3]
    rgb_o, d_o = fo.rasterize(v, f, None, poses, 0.25, 600, 600, 210, 210, 420, 420)
    rgb_g, d_g = ops.rasterize(ops.Mesh(v, f, None), torch.from_numpy(poses), 0.25, 600, 600, 210, 210, 420, 420)
    assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32))
    assert np.array_equal(rgb_g.cpu().numpy(), rgb_o)
    assert (rgb_o[d_o > 0] == 255).all()
    # the nearer small triangle occludes the quad at the image centre
    assert abs(d_o[0, 210, 210] - (1.1 - 0.075)) < 1e-3


@pytest.mark.parametrize("sub,W,H", [(2, 420, 420), (4, 518, 518), (3, 300, 200), (5, 704, 480)])
def test_rasterizer_tiled_and_global_paths_are_bit_identical(sub, W, H):
    """both visibility strategies (LDS tiles with chunk-mask binning; global 64-bit atomic buffer) must give the oracle's
    image: tile edges 64 / 72 / 88 px, non-square frames, big triangles (two quads in front), poses partly off-screen"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    v, f = _icosphere(sub)
    v = v * (1 + 0.2 * np.sin(4 * v[:, :1]))
    n0 = len(v)
    v = np.concatenate([v, np.array([[-1.5, -1.5, 1.2], [1.5, -1.5, 1.2], [1.5, 1.5, 1.2], [-1.5, 1.5, 1.2]], v.dtype)])
    f = np.concatenate([f, np.array([[n0, n0 + 1, n0 + 2], [n0, n0 + 2, n0 + 3]], f.dtype)])     # a big back-plane quad
    colors = np.random.Generator(np.random.PCG64(5)).integers(0, 256, size=(len(v), 3), dtype=np.uint8)
    poses = np.tile(np.eye(4, dtype=np.float32), (4, 1, 1))
    poses[:, :3, :3] = fo.generate_rotations(4)
    poses[:, :3, 3] = [[0, 0, 1.1], [0.25, -0.1, 0.9], [-0.4, 0.3, 1.4], [0, 0, 0.6]]
    fx = 600.0 * W / 420
    rgb_o, d_o = fo.rasterize(v, f, colors, poses, 0.25, fx, fx, W / 2, H / 2, W, H)
    mesh = ops.Mesh(v, f, colors)
    try:
        for mode in (1, 0):
            ops.set_option("raster_tiled", mode)
            rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(poses), 0.25, fx, fx, W / 2, H / 2, W, H)
            assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32)), f"depth differs (tiled={mode})"
            assert np.array_equal(rgb_g.cpu().numpy(), rgb_o), f"rgb differs (tiled={mode})"
    finally:
        ops.set_option("raster_tiled", -1)
    assert (d_o > 0).mean() > 0.3


def test_rerank_views_bit_exact_and_bank_api():
    """per-view fine re-rank (--topk 25 path): HIP kernel vs oracle bit for bit; TemplateBank.retrieve_reranked picks the
    planted mesh and resolves ties to the first maximum in coarse order"""
    from freepose_amd import ops
    from freepose_amd.retrieval import TemplateBank
    from oracle import fp_oracle as fo
    from tests.test_golden_cpu import _rerank_case
    for D in (1024, 384):
        views, counts, q, cand = _rerank_case(D=D)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        view_bits = fo.to_bf16_bits(np.concatenate(views))
        q_bits = fo.l2norm_rows(fo.to_bf16_bits(q))
        for k in (1, 7, 25, 64):
            o = fo.rerank_views(view_bits, off, cand, q_bits, k)
            g = ops.rerank_views(fo.bits_to_torch(view_bits), torch.from_numpy(off), torch.from_numpy(cand), fo.bits_to_torch(q_bits), k)
            assert np.array_equal(g.cpu().numpy().view(np.uint32), o.view(np.uint32)), (D, k)
    views, counts, q, cand = _rerank_case()
    bank = TemplateBank(np.stack([v.mean(axis=0) for v in views]), [f"m{i}" for i in range(len(views))])
    bank.attach_views(views)
    names, scores, rows, fine = bank.retrieve_reranked(ops.l2_normalize(torch.from_numpy(q).to(torch.bfloat16)), topk=25, n_coarse=12)
    assert names == ["m3", "m8"] and fine.shape == (2, 12)
