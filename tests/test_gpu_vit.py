"""End-to-end ViT parity on the MI355X: fp_vit_forward (HIP, bf16 storage / fp32 accumulate) vs the torch-fp32
CPU restatement oracle/vit_ref.py on identical seeded weights and images.

Tolerance (stated here): per-patch cosine >= 0.999 and relative L2 <= 2e-2 against the fp32
oracle — the same distance class the reference's own bf16 model (oracle run in bf16) shows against fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(B, H, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    # smooth-ish synthetic crops in [0,1] (pure white noise would be a degenerate ViT input)
    low = rng.random((B, 3, H // 14, H // 14)).astype(np.float32)
    img = torch.nn.functional.interpolate(torch.from_numpy(low), size=(H, H), mode="bilinear")
    img = 0.8 * img + 0.2 * torch.from_numpy(rng.random((B, 3, H, H)).astype(np.float32))
    return img.to(torch.bfloat16)


def _metrics(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=-1)
    rel = ((got - ref).norm() / ref.norm()).item()
    return cos.min().item(), rel


@pytest.mark.parametrize("model,H,layer,B", [("dinov2_vits14_reg", 224, 22, 2), ("dinov2_vitl14_reg", 420, 22, 2),
                                             ("dinov2_vitl14_reg", 518, 2, 1), ("dinov2_vits14_reg", 224, 5, 3),
                                             ("dinov2_vitl14_reg", 518, 22, 1),      # the bench / north-star configuration
                                             ("dinov2_vitb14_reg", 518, 12, 1)])     # TrackingRefiner
def test_vit_forward_vs_fp32_oracle(model, H, layer, B):
    from freepose_amd import ops
    from oracle import vit_ref
    sd = ops.random_state_dict(model, seed=3)
    img = _images(B, H, 7)
    vit = ops.ViT(model, sd)
    sdf = {k: v.float() for k, v in sd.items()}
    for ft in ("patch", "cls", "reg"):
        if ft == "reg" and vit.n_reg == 0:
            continue
        got = vit(img, layer=layer, feature_type=ft)
        torch.cuda.synchronize()
        ref = vit_ref.vit_forward(sdf, img.float(), layer=layer, feature_type=ft, dtype=torch.float32)
        assert got.shape == ref.shape
        cmin, rel = _metrics(got, ref)
        print(f"{model} {H} layer={layer} {ft}: min cos {cmin:.5f} rel {rel:.4f}")
        assert cmin >= 0.999 and rel <= 2e-2, (model, ft, cmin, rel)
    if model == "dinov2_vits14_reg":
        # distance of the reference's own precision regime (torch CPU bf16 model) to fp32, for context
        ref32 = vit_ref.vit_forward(sdf, img.float(), layer=layer, feature_type="patch", dtype=torch.float32)
        ref16 = vit_ref.vit_forward(sdf, img.float(), layer=layer, feature_type="patch", dtype=torch.bfloat16)
        c16, r16 = _metrics(ref16, ref32)
        got = vit(img, layer=layer, feature_type="patch")
        cg, rg = _metrics(got, ref32)
        print(f"torch-bf16 vs fp32: cos {c16:.5f} rel {r16:.4f} | hip vs fp32: cos {cg:.5f} rel {rg:.4f}")
        assert rg <= 2.0 * r16 + 2e-3


def _massive_state_dict(model, seed):
    """random-init weights bent towards what trained DINOv2-reg checkpoints look like (parity caveat i): a few residual
    channels carry activations hundreds of times the typical magnitude at the CLS / register tokens and at a handful of patch
    positions, LayerNorm gains span an order of magnitude with non-zero shifts, LayerScale gammas are small and uneven."""
    from freepose_amd import ops
    sd = {k: v.float() for k, v in ops.random_state_dict(model, seed=seed).items()}
    g = torch.Generator().manual_seed(seed + 100)
    dim = sd["norm.weight"].numel()
    big = [5, dim // 3, dim - 7]
    sd["cls_token"][..., big[0]] += 90.0
    if "register_tokens" in sd:
        sd["register_tokens"][..., big[1]] += 160.0
        sd["register_tokens"][:, :2, big[2]] -= 70.0
    sd["pos_embed"][:, 40:44, big[1]] += 220.0            # a few patch tokens with a massive channel
    sd["pos_embed"][:, 300:302, big[0]] -= 130.0
    for k in list(sd):
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
            sd[k] = torch.exp(torch.empty(dim).uniform_(-1.2, 1.0, generator=g))
        elif k.endswith("norm1.bias") or k.endswith("norm2.bias") or k == "norm.bias":
            sd[k] = torch.empty(dim).uniform_(-0.5, 0.5, generator=g)
        elif k.endswith("gamma"):
            sd[k] = torch.exp(torch.empty(dim).uniform_(-4.0, 0.0, generator=g))
    return {k: v.to(torch.bfloat16) for k, v in sd.items()}


@pytest.mark.parametrize("model,H,layer,B", [("dinov2_vits14_reg", 224, 22, 2), ("dinov2_vitl14_reg", 420, 22, 1)])
def test_vit_massive_activation_regime(model, H, layer, B):
    """Seeded weights with massive-activation channels, wide LayerNorm gains and small LayerScale (what real checkpoints have and
    plain random init does not).  Stated bar: against the fp32 oracle the HIP forward — with the LayerNorm folded into the GEMMs and
    with the separate LayerNorm kernel — is no further away than twice the distance of the reference's own regime (the torch bf16
    model, dino.py:14 / pose_estimator.py:21) plus 2e-3, per-patch cosine >= 0.995."""
    from freepose_amd import ops
    from oracle import vit_ref
    sd = _massive_state_dict(model, 5)
    img = _images(B, H, 11)
    sdf = {k: v.float() for k, v in sd.items()}
    ref32 = vit_ref.vit_forward(sdf, img.float(), layer=layer, feature_type="patch", dtype=torch.float32)
    ref16 = vit_ref.vit_forward(sd, img.float(), layer=layer, feature_type="patch", dtype=torch.bfloat16)
    c16, r16 = _metrics(ref16, ref32)
    # the regime is really there: the residual stream the last block sees has channels far above its typical magnitude
    x = vit_ref.vit_forward(sdf, img.float(), layer=max(layer - 1, 1), feature_type="patch", dtype=torch.float32)
    vit = ops.ViT(model, sd)
    try:
        for fused in (1, 0):
            ops.set_option("ln_fused", fused)
            got = vit(img, layer=layer, feature_type="patch")
            torch.cuda.synchronize()
            cg, rg = _metrics(got, ref32)
            print(f"{model} massive-activation regime, ln_fused={fused}: hip vs fp32 cos {cg:.5f} rel {rg:.4f} | torch-bf16 vs fp32 cos {c16:.5f} rel {r16:.4f}")
            assert torch.isfinite(got.float()).all()
            assert cg >= 0.995 and rg <= 2.0 * r16 + 2e-3, (model, fused, cg, rg, c16, r16)
    finally:
        ops.set_option("ln_fused", -1)
    assert float(x.abs().max()) > 20.0 * float(x.abs().median())


def test_vit_batch_invariance_and_layer_semantics():
    from freepose_amd import ops
    vit = ops.ViT("dinov2_vits14_reg", seed=1)
    img = _images(5, 224, 9)
    full = vit(img, layer=22, feature_type="patch")
    one = vit(img[2:3], layer=22, feature_type="patch")
    assert torch.equal(full[2:3], one), "a crop's features must not depend on its batch neighbours"
    # layer > depth runs every block (dino.py:18-21) == layer == depth
    assert torch.equal(vit(img[:1], layer=12, feature_type="cls"), vit(img[:1], layer=99, feature_type="cls"))
    assert not torch.equal(vit(img[:1], layer=11, feature_type="cls"), vit(img[:1], layer=12, feature_type="cls"))


def test_checkpoint_path_hub_layout_fp32_pth(tmp_path, monkeypatch):
    """dino.py:8-12 loads `dinov2_vitl14_reg4_pretrain.pth` through torch.hub; here the same file is found through
    FREEPOSE_DINOV2_WEIGHTS (file or directory).  A hub-layout fp32 checkpoint (incl. `mask_token`, which the forward never
    uses) must give exactly the features of the in-memory load of the same weights; a missing file must fail closed."""
    from freepose_amd import ops
    from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor, _CKPT_NAMES
    name = "dinov2_vits14_reg"
    sd = ops.random_state_dict(name, seed=11)
    hub = {k: v.float().clone() for k, v in sd.items()}            # the hub files are fp32
    hub["mask_token"] = torch.zeros(1, 384)
    ck = tmp_path / _CKPT_NAMES[name]
    torch.save(hub, ck)
    x = torch.rand(3, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).cuda()
    ref = DINOv2FeatureExtractor(name, state_dict=sd)(x, layer=12, feature_type="patch")
    for env in (str(ck), str(tmp_path)):                            # file and directory forms
        monkeypatch.setenv("FREEPOSE_DINOV2_WEIGHTS", env)
        fe = DINOv2FeatureExtractor(name)
        assert fe.checkpoint == ck
        got = fe(x, layer=12, feature_type="patch")
        assert torch.equal(got, ref)
        assert torch.equal(fe(x, layer=12, feature_type="cls"), DINOv2FeatureExtractor(name, state_dict=sd)(x, layer=12, feature_type="cls"))
    monkeypatch.setenv("FREEPOSE_DINOV2_WEIGHTS", str(tmp_path / "nowhere"))
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path / "empty_hub"))
    monkeypatch.delenv("FREEPOSE_ALLOW_RANDOM_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError, match="FREEPOSE_DINOV2_WEIGHTS"):
        DINOv2FeatureExtractor(name)
    with pytest.warns(RuntimeWarning):
        DINOv2FeatureExtractor(name, allow_random_weights=True)


@pytest.mark.parametrize("model,H,B", [("dinov2_vits14_reg", 224, 3), ("dinov2_vitl14_reg", 518, 2), ("dinov2_vitb14_reg", 420, 1)])
def test_patch_normalized_equals_normalizing_the_patch_features(model, H, B):
    """feature_type 'patch_normalized' (round 6): the final-norm kernel writes F.normalize()d rows itself (pose_estimator.py:85-88
    normalises the hypothesis features right after the ViT) — the bits of fp_l2_normalize on the plain patch features, and scoring with
    the streaming kernel gives the on-the-fly kernel's bits"""
    from freepose_amd import ops
    vit = ops.ViT(model, ops.random_state_dict(model, seed=6))
    img = _images(B, H, 9)
    plain = vit(img, layer=22, feature_type="patch")
    fused = vit(img, layer=22, feature_type="patch_normalized")
    assert torch.equal(fused.view(torch.int16), ops.l2_normalize(plain).view(torch.int16))
    q = ops.l2_normalize(plain[0])
    assert torch.equal(ops.template_score(fused, q, normalized=True), ops.template_score(plain, q))
