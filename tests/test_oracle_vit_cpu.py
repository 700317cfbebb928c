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
