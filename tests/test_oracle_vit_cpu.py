"""Pins oracle/vit_ref.py (the CPU restatement of the un-vendored hub DINOv2 forward) against an independent
implementation available in the image: transformers' Dinov2WithRegistersModel with the same weights (SURVEY App. B key
map), driven layer-by-layer like src/pipeline/retrieval/dino.py:16-30.  Also checks the `layer` semantics and the
pos-embed interpolation.  No GPU."""
import pytest
import torch

from oracle import vit_ref


def _sd(model, seed):
    from freepose_amd.ops import random_state_dict
    return {k: v.float() for k, v in random_state_dict(model, seed).items()}


def _hf(sd, dim, depth, heads, n_reg):
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    cfg = Dinov2WithRegistersConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, patch_size=14,
                                    image_size=518, num_register_tokens=n_reg, layerscale_value=1.0, mlp_ratio=4)
    m = Dinov2WithRegistersModel(cfg).eval()
    missing = m.load_state_dict(vit_ref.to_hf_state_dict(sd), strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    return m


def _hf_forward(m, x, layer):
    with torch.no_grad():
        h = m.embeddings(vit_ref.normalize_images(x))
        for i, blk in enumerate(m.encoder.layer):
            h = blk(h)
            h = h[0] if isinstance(h, tuple) else h
            if i + 1 == layer:
                break
        return m.layernorm(h)


@pytest.mark.parametrize("res,layer", [(224, 22), (420, 5), (518, 3)])
def test_vit_ref_matches_transformers(res, layer):
    sd = _sd("dinov2_vits14_reg", 3)
    m = _hf(sd, 384, 12, 6, 4)
    x = torch.rand(2, 3, res, res, generator=torch.Generator().manual_seed(1))
    ref = _hf_forward(m, x, layer)
    for ft, sl in (("cls", lambda h: h[:, 0]), ("reg", lambda h: h[:, 1:5]), ("patch", lambda h: h[:, 5:])):
        mine = vit_ref.vit_forward(sd, x, layer=layer, feature_type=ft)
        assert mine.shape == sl(ref).shape
        assert (mine - sl(ref)).abs().max().item() <= 1e-4 * sl(ref).abs().max().item()


def test_layer_beyond_depth_runs_all_blocks():
    sd = _sd("dinov2_vits14_reg", 4)
    x = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    a = vit_ref.vit_forward(sd, x, layer=22, feature_type="cls")   # 12-block model: never breaks (dino.py:18-21)
    b = vit_ref.vit_forward(sd, x, layer=12, feature_type="cls")
    c = vit_ref.vit_forward(sd, x, layer=11, feature_type="cls")
    assert torch.equal(a, b) and not torch.equal(b, c)


def test_pos_embed_interpolation_shapes_and_identity():
    pe = torch.randn(1, 1 + 37 * 37, 64, generator=torch.Generator().manual_seed(3))
    assert vit_ref.interpolate_pos_encoding(pe, 37, 37) is pe            # 518 px: native grid, untouched
    out = vit_ref.interpolate_pos_encoding(pe, 30, 30)                   # 420 px
    assert out.shape == (1, 901, 64) and torch.equal(out[:, 0], pe[:, 0])
    out16 = vit_ref.interpolate_pos_encoding(pe, 16, 16)                 # 224 px
    assert out16.shape == (1, 257, 64)


def _truncate(sd, depth):
    """first `depth` blocks of a hub-layout state dict (a shallower model with the same widths)"""
    return {k: v for k, v in sd.items() if not k.startswith("blocks.") or int(k.split(".")[1]) < depth}


@pytest.mark.parametrize("model,dim,heads,res", [("dinov2_vitl14_reg", 1024, 16, 518), ("dinov2_vitl14_reg", 1024, 16, 420),
                                                 ("dinov2_vitb14_reg", 768, 12, 518), ("dinov2_vitb14_reg", 768, 12, 420)])
def test_vit_ref_matches_transformers_at_vitl_and_vitb_widths(model, dim, heads, res):
    """the restatement at the widths the pipeline uses (ViT-L/14-reg: retrieval + pose, ViT-B/14-reg: TrackingRefiner), two
    blocks deep, at both resolutions (518: native 37 x 37 pos-embed grid, 420: bicubic-antialias resize to 30 x 30), with
    non-trivial LayerScale so that ls1 / ls2 are not interchangeable"""
    sd = _truncate(_sd(model, 5), 2)
    g = torch.Generator().manual_seed(11)
    for i in range(2):
        sd[f"blocks.{i}.ls1.gamma"] = 0.5 + torch.rand(dim, generator=g)
        sd[f"blocks.{i}.ls2.gamma"] = 0.5 + torch.rand(dim, generator=g)
    m = _hf(sd, dim, 2, heads, 4)
    x = torch.rand(1, 3, res, res, generator=torch.Generator().manual_seed(7))
    ref = _hf_forward(m, x, 22)                       # layer > depth: all (two) blocks, then the final norm
    mine = vit_ref.vit_forward(sd, x, layer=22, feature_type="patch")
    P = (res // 14) ** 2
    assert mine.shape == (1, P, dim) and ref.shape == (1, P + 5, dim)
    rel = (mine - ref[:, 5:]).abs().max().item() / ref[:, 5:].abs().max().item()
    assert rel <= 1e-4, rel                           # fp32 summation-order noise only (measured ~1e-6)
    cls = vit_ref.vit_forward(sd, x, layer=1, feature_type="cls")
    assert (cls - _hf_forward(m, x, 1)[:, 0]).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("res", [420, 518])
def test_vit_ref_matches_transformers_full_depth_vitl(res):
    """the bench / pipeline configuration itself: ViT-L/14-reg, the first 22 of 24 blocks + final norm (dino.py:18-23, layer = 22),
    at 420^2 (pose estimators) and 518^2 (BASELINE metric), non-trivial LayerScale — the oracle restatement against transformers'
    Dinov2WithRegistersModel with the same weights, all 22 blocks deep"""
    sd = _sd("dinov2_vitl14_reg", 9)
    g = torch.Generator().manual_seed(13)
    for i in range(24):
        sd[f"blocks.{i}.ls1.gamma"] = 0.05 + 0.3 * torch.rand(1024, generator=g)
        sd[f"blocks.{i}.ls2.gamma"] = 0.05 + 0.3 * torch.rand(1024, generator=g)
    m = _hf(sd, 1024, 24, 16, 4)
    x = torch.rand(1, 3, res, res, generator=torch.Generator().manual_seed(3))
    ref = _hf_forward(m, x, 22)
    mine = vit_ref.vit_forward(sd, x, layer=22, feature_type="patch")
    assert mine.shape == ref[:, 5:].shape == (1, (res // 14) ** 2, 1024)
    rel = (mine - ref[:, 5:]).abs().max().item() / ref[:, 5:].abs().max().item()
    assert rel <= 1e-4, rel
