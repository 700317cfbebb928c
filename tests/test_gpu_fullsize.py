"""Full-size (BASELINE.json) cases on the MI355X, checked through size-independent properties: the oracle cannot run these
sizes in seconds, so each test ties the full-size path to something that IS verified against the oracle at small size
(bit-exact equality with the small-tile kernels, determinism, canonical ordering, a planted known answer)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(B, H, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((B, 3, H, H), generator=g).to(torch.bfloat16)


def test_vit_l_518_big_batch_equals_small_batches_bitwise():
    """64 crops @518^2 run every linear layer through the persistent 256x256 kernels (qk / v / proj / fc1 / fc2 and the
    patch embed, streaming row-coalesced epilogues); the same crops two at a time use the 128x128 kernels that
    tests/test_gpu_vit.py checks against the fp32 oracle.  Per-row K order is identical in both, so the features must be
    BIT-identical — any tile-walk, epilogue-slab or transposed-store indexing error shows up as a mismatch."""
    from freepose_amd import ops
    vit = ops.ViT("dinov2_vitl14_reg", seed=3)
    img = _images(64, 518, 17).cuda()
    big = vit(img, layer=22, feature_type="patch")
    assert big.shape == (64, 1369, 1024) and torch.isfinite(big.float()).all()
    for i in (0, 31, 62):
        small = vit(img[i:i + 2], layer=22, feature_type="patch")
        assert torch.equal(big[i:i + 2], small), f"crops {i},{i + 1}: big-batch features differ from the small-batch path"
    # one crop alone: the 64x64 tier on its deepest K-tile ring, row statistics finalised in the qk / fc1 prologues (big batch: the
    # finalisation kernel) — still the same bits
    assert torch.equal(big[5:6], vit(img[5:6], layer=22, feature_type="patch")), "single-crop features differ from the big-batch path"
    again = vit(img, layer=22, feature_type="patch")
    assert torch.equal(big, again), "the forward must be deterministic"


def test_full_bank_topk_invariants():
    """46 037 x 1024 bank: one query per pass, four per pass and sixteen per call must give identical (score, index) lists;
    lists are sorted by the canonical rule (score desc, index asc), indices unique and in range; a planted row is top-1;
    two bank shards merged equal the unsharded result."""
    import bench
    from freepose_amd import ops
    from freepose_amd.retrieval import TemplateBank
    bank_f32 = bench.synthetic_bank(46037, 1024, seed=21)
    tb = TemplateBank(bank_f32, shard=False)
    rng = np.random.default_rng(5)
    q = rng.standard_normal((16, 1024)).astype(np.float32)
    q[3] = bank_f32[31337] * 3.0                      # planted: parallel to bank row 31337
    qd = ops.l2_normalize(torch.from_numpy(q).cuda().to(torch.bfloat16))
    s16, i16 = tb.topk(qd, 100)
    s16, i16 = s16.cpu().numpy(), i16.cpu().numpy()
    assert i16[3, 0] == 31337
    for r in range(16):
        s1, i1 = tb.topk(qd[r:r + 1], 100)
        assert np.array_equal(s1.cpu().numpy()[0], s16[r]) and np.array_equal(i1.cpu().numpy()[0], i16[r])
        assert len(set(i16[r].tolist())) == 100 and i16[r].min() >= 0 and i16[r].max() < 46037
        ds = np.diff(s16[r])
        assert (ds <= 0).all()
        ties = np.where(ds == 0)[0]
        assert (i16[r][ties] < i16[r][ties + 1]).all(), "equal scores must be ordered by ascending index"
    s4, i4 = tb.topk(qd[4:8], 100)
    assert np.array_equal(i4.cpu().numpy(), i16[4:8]) and np.array_equal(s4.cpu().numpy(), s16[4:8])
    # two shards + merge == unsharded
    half = 23000
    a, b = TemplateBank(bank_f32[:half], shard=False), TemplateBank(bank_f32[half:], shard=False)
    sa, ia = a.topk(qd, 100)
    sb, ib = b.topk(qd, 100)
    ms, mi = ops.topk_merge(torch.cat([sa, sb], 1), torch.cat([ia, ib + half], 1).to(torch.int32), 100)
    assert np.array_equal(mi.cpu().numpy(), i16) and np.array_equal(ms.cpu().numpy(), s16)


def test_hot_path_full_size_planted_hypothesis():
    """BASELINE config 3 sizes: 576 hypotheses of an 81 920-triangle mesh at 420^2, 518^2 crops, ViT-L.  The query is the
    crop of hypothesis 123's own render, so the render-and-compare stage must return 123 with a near-1 score, and the
    metric depth must follow the reference's extents formula."""
    import bench
    from freepose_amd import ops
    from freepose_amd.pipeline import HotPath
    from freepose_amd.retrieval import TemplateBank
    vit = ops.ViT("dinov2_vitl14_reg", seed=0)
    bank = TemplateBank(bench.synthetic_bank(4096, 1024, seed=2), shard=False)
    v, f, c = bench.synthetic_mesh(6)
    hp = HotPath(vit, bank, ops.Mesh(v, f, c), n_hyp=576, crop_res=518)
    hyp_crops, ext = hp.render_hypotheses()
    assert hyp_crops.shape == (576, 3, 518, 518)
    j = 123
    masks = (hyp_crops[j:j + 1].float().amax(1) > 0)
    K = np.array([[hp.fx, 0, hp.cx], [0, hp.fy, hp.cy], [0, 0, 1]], dtype=np.float64)
    e = ext[j].cpu().numpy()
    bbox = np.array([[e[0], e[1], e[2], e[3]]])          # xyxy, as z_from_extents expects
    res = hp.run(hyp_crops[j:j + 1], masks, K, bbox, [hp.render_scale])
    assert res[0].hyp_idx[0] == j and res[0].hyp_scores[0] > 0.99
    # get_z_from_pointcloud (src/pipeline/utils.py) estimates the depth of the SILHOUETTE points: for the 0.25-radius mesh
    # rendered at z = 1.1 that is z - r^2/z = 1.043 .. 1.06 (pixel quantisation), and it must equal the formula on the extents
    z = res[0].TCO[0][2, 3]
    z_formula = (hp.fy * e[5] / (e[3] - e[1] + 1) + hp.fx * e[4] / (e[2] - e[0] + 1)) / 2
    assert abs(z - z_formula) < 1e-9 and 1.03 < z < 1.08
    res2 = hp.run(hyp_crops[j:j + 1], masks, K, bbox, [hp.render_scale])
    assert np.array_equal(res[0].hyp_scores, res2[0].hyp_scores) and np.array_equal(res[0].topk_idx, res2[0].topk_idx)


def test_bank_build_config_at_stated_size():
    """BASELINE config 2 at its stated size: ViT-L/14-reg layer-22 FFA with batch 256 over the 42 template views of an object
    (scripts.extract_retrieval_features.mesh_descriptors, reference :36-70).  The 256-crop call is padded out with repeats of
    the views; every crop's features must be bit-identical to the same crop in a 2-crop call (batch invariance at B = 256 @
    420^2), and the object descriptor must equal the mean of the per-view descriptors computed one view at a time."""
    import warnings

    from freepose_amd import ops
    from freepose_amd.scripts.extract_retrieval_features import mesh_descriptors
    from src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = DINOv2FeatureExtractor("dinov2_vitl14_reg", seed=1)
    g = torch.Generator().manual_seed(10)
    views = torch.rand(42, 3, 420, 420, generator=g).cuda()
    yy, xx = torch.meshgrid(torch.arange(420), torch.arange(420), indexing="ij")
    masks = torch.stack([(((yy - 210 + 3 * i) / (90.0 + i)) ** 2 + ((xx - 200 - 2 * i) / (150.0 - 2 * i)) ** 2) <= 1 for i in range(42)]).cuda()
    big = torch.cat([views] * 7)[:256]                                    # B = 256
    feats = model(big, layer=22, feature_type="patch")
    assert feats.shape == (256, 900, 1024)
    pair = model(views[:2], layer=22, feature_type="patch")
    assert torch.equal(feats[:2], pair) and torch.equal(feats[42:44], pair) and torch.equal(feats[252:254], feats[0:2])
    desc = mesh_descriptors(model, {"templates": views, "masks": masks}, "ffa", 22, 256)
    assert desc.shape == (42, 1024) and desc.dtype == np.float32 and np.isfinite(desc).all()
    one = np.concatenate([mesh_descriptors(model, {"templates": views[i:i + 1], "masks": masks[i:i + 1]}, "ffa", 22, 256) for i in (0, 17, 41)])
    assert np.array_equal(desc[[0, 17, 41]], one)
