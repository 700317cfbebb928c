"""CPU restatement (plain torch ops) of the DINOv2 ViT forward that the reference drives through torch.hub.

TEST INFRASTRUCTURE ONLY — never imported by freepose_amd/.

What it restates: `DINOv2FeatureExtractor.forward` (src/pipeline/retrieval/dino.py:14-32) and, because
`facebookresearch/dinov2` is un-vendored / unpinned in /root/reference (dino.py:10; environment_cuda.yaml pins
only torch/transformers/timm), the published DINOv2 `vit_large(patch_size=14, num_register_tokens=4,
init_values=1.0, interpolate_antialias=True, interpolate_offset=0.0)` algorithm:
  prepare_tokens_with_masks: Conv2d(3,D,14,14) patch embed -> [cls | patches] + bicubic/antialias-resized
  pos-embed -> registers inserted after cls; blocks: x += ls1*proj(attn(norm1 x)); x += ls2*fc2(gelu(fc1(norm2 x))).
Parity of this restatement is pinned against transformers' Dinov2WithRegistersModel (tests/test_oracle_vit.py,
key map of SURVEY App. B); parity vs the un-vendored hub code itself is UNPINNED (no network, no hub cache).

dtype=torch.float32 is the numerical reference for the HIP kernels; dtype=torch.bfloat16 reproduces the
reference's own rounding points (whole model cast to bf16, pose_estimator.py:21).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def interpolate_pos_encoding(pos_embed: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    """pos_embed [1, 1+G*G, D] -> [1, 1+gh*gw, D] (DINOv2 interpolate_pos_encoding, offset 0, antialias)."""
    prev = pos_embed.dtype
    N = pos_embed.shape[1] - 1
    G = int(math.isqrt(N))
    if gh * gw == N and gh == gw:
        return pos_embed
    pe = pos_embed.float()
    cls_pos, patch_pos = pe[:, :1], pe[:, 1:]
    D = pe.shape[-1]
    patch_pos = patch_pos.reshape(1, G, G, D).permute(0, 3, 1, 2)
    patch_pos = F.interpolate(patch_pos, size=(gh, gw), mode="bicubic", antialias=True)
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, gh * gw, D)
    return torch.cat([cls_pos, patch_pos], dim=1).to(prev)


def normalize_images(images: torch.Tensor) -> torch.Tensor:
    """torchvision.transforms.Normalize on the image dtype (dino.py:12,16): sub_ then div_."""
    mean = torch.as_tensor(IMAGENET_MEAN, dtype=images.dtype).view(1, 3, 1, 1)
    std = torch.as_tensor(IMAGENET_STD, dtype=images.dtype).view(1, 3, 1, 1)
    return (images - mean) / std


def attention(x, w_qkv, b_qkv, w_proj, b_proj, heads):
    B, N, D = x.shape
    qkv = F.linear(x, w_qkv, b_qkv).reshape(B, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    o = F.scaled_dot_product_attention(q, k, v)  # softmax(q k^T / sqrt(hd)) v
    o = o.transpose(1, 2).reshape(B, N, D)
    return F.linear(o, w_proj, b_proj)


def vit_forward(sd: dict, images: torch.Tensor, layer: int = 22, feature_type: str = "patch", heads: int | None = None,
                n_reg: int | None = None, dtype=torch.float32, eps: float = 1e-6) -> torch.Tensor:
    """sd: hub-layout state dict; images [B,3,H,W] in [0,1].  Returns cls [B,D] / reg [B,R,D] / patch [B,P,D]."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    x = normalize_images(images.to(dtype))
    D = sd["cls_token"].shape[-1]
    heads = heads or D // 64
    n_reg = sd["register_tokens"].shape[1] if (n_reg is None and "register_tokens" in sd) else (n_reg or 0)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    B, _, H, W = x.shape
    gh, gw = H // 14, W // 14
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=14)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + interpolate_pos_encoding(sd["pos_embed"], gh, gw)
    if n_reg:
        x = torch.cat([x[:, :1], sd["register_tokens"].expand(B, -1, -1), x[:, 1:]], dim=1)
    for i in range(depth):
        p = f"blocks.{i}."
        y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        y = attention(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], heads)
        x = x + (y * sd[p + "ls1.gamma"] if p + "ls1.gamma" in sd else y)
        y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = x + (y * sd[p + "ls2.gamma"] if p + "ls2.gamma" in sd else y)
        if i + 1 == layer:  # dino.py:18-21 — never true when layer > depth
            break
    x = F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], eps)
    if feature_type == "cls":
        return x[:, 0]
    if feature_type == "reg":
        return x[:, 1:1 + n_reg]
    return x[:, 1 + n_reg:]


def vit_forward_video_regime(sd: dict, images: torch.Tensor, layer: int = 22, feature_type: str = "patch", eps: float = 1e-6) -> torch.Tensor:
    """The VIDEO script's precision regime, restated op by op: a bf16 model (online_pose_estimator.py:19) fed fp16 images
    (`.half()`, online_pose_estimator.py:51,66) under `torch.autocast(device_type='cuda', dtype=torch.bfloat16)`
    (scripts/dino_inference_video.py:151).  CUDA autocast semantics [torch/csrc/autocast_mode.cpp; public knowledge]:
      * conv / linear / matmul / scaled-dot-product attention cast their floating inputs to bf16 and return bf16;
      * layer_norm and softmax run in fp32 and RETURN fp32 (so the final norm's features are fp32, not bf16);
      * elementwise ops (Normalize's sub / div, GELU, LayerScale mul, residual add) run in their input dtype with ordinary
        type promotion (bf16 op fp32 -> fp32).
    images: float32 tensor holding the values of the fp16 input (any float input is rounded to fp16 first).  Returns fp32 features."""
    bf, h = torch.bfloat16, torch.float16
    sd = {k: v.to(bf) for k, v in sd.items()}
    mean = torch.as_tensor(IMAGENET_MEAN, dtype=h).view(1, 3, 1, 1)
    std = torch.as_tensor(IMAGENET_STD, dtype=h).view(1, 3, 1, 1)
    x = (images.to(h) - mean) / std                                        # T.Normalize on the fp16 tensor (fp16 arithmetic)
    D = sd["cls_token"].shape[-1]
    heads = D // 64
    n_reg = sd["register_tokens"].shape[1] if "register_tokens" in sd else 0
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    B, _, H, W = x.shape
    gh, gw = H // 14, W // 14
    x = F.conv2d(x.to(bf), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=14)     # autocast: conv in bf16
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + interpolate_pos_encoding(sd["pos_embed"], gh, gw)
    if n_reg:
        x = torch.cat([x[:, :1], sd["register_tokens"].expand(B, -1, -1), x[:, 1:]], dim=1)

    def ln(t, w, b):                                                        # autocast: fp32 in, fp32 out
        return F.layer_norm(t.float(), (D,), w.float(), b.float(), eps)

    def lin(t, w, b):                                                       # autocast: bf16 in, bf16 out
        return F.linear(t.to(bf), w, b)

    for i in range(depth):
        p = f"blocks.{i}."
        y = ln(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        qkv = lin(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, -1, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, -1, D)
        y = lin(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        x = x + (y * sd[p + "ls1.gamma"] if p + "ls1.gamma" in sd else y)
        y = ln(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        y = lin(F.gelu(lin(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = x + (y * sd[p + "ls2.gamma"] if p + "ls2.gamma" in sd else y)
        if i + 1 == layer:
            break
    x = ln(x, sd["norm.weight"], sd["norm.bias"])                           # fp32 features leave the extractor
    if feature_type == "cls":
        return x[:, 0]
    if feature_type == "reg":
        return x[:, 1:1 + n_reg]
    return x[:, 1 + n_reg:]


def score_video_regime(query_feat_f32: torch.Tensor, template_feats_f32: torch.Tensor) -> torch.Tensor:
    """online_pose_estimator.py:52,76 under the same autocast: F.normalize on the fp32 features (fp32), einsum -> bmm in bf16 (inputs
    rounded to bf16, fp32 accumulation, bf16 result), .mean(-1) on the bf16 dots.  query [P,D], templates [T,P,D] -> [T] (bf16 values)."""
    q = F.normalize(query_feat_f32, dim=-1).to(torch.bfloat16)
    t = F.normalize(template_feats_f32, dim=-1).to(torch.bfloat16)
    dots = (t.float() * q.float()[None]).sum(-1).to(torch.bfloat16)        # 'b n d, b n d -> b n' as a batched matmul in bf16
    return dots.float().mean(-1).to(torch.bfloat16).float()


def to_hf_state_dict(sd: dict) -> dict:
    """hub DINOv2 names -> transformers Dinov2WithRegistersModel names (SURVEY.md App. B key map)."""
    D = sd["cls_token"].shape[-1]
    out = {"embeddings.cls_token": sd["cls_token"], "embeddings.position_embeddings": sd["pos_embed"],
           "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
           "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
           "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"],
           "embeddings.mask_token": torch.zeros(1, D)}
    if "register_tokens" in sd:
        out["embeddings.register_tokens"] = sd["register_tokens"]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(depth):
        p, h = f"blocks.{i}.", f"encoder.layer.{i}."
        for n in ("norm1", "norm2"):
            out[h + n + ".weight"], out[h + n + ".bias"] = sd[p + n + ".weight"], sd[p + n + ".bias"]
        w, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, nm in enumerate(("query", "key", "value")):
            out[h + f"attention.attention.{nm}.weight"] = w[j * D:(j + 1) * D]
            out[h + f"attention.attention.{nm}.bias"] = b[j * D:(j + 1) * D]
        out[h + "attention.output.dense.weight"], out[h + "attention.output.dense.bias"] = sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]
        out[h + "layer_scale1.lambda1"], out[h + "layer_scale2.lambda1"] = sd[p + "ls1.gamma"], sd[p + "ls2.gamma"]
        for n in ("fc1", "fc2"):
            out[h + f"mlp.{n}.weight"], out[h + f"mlp.{n}.bias"] = sd[p + f"mlp.{n}.weight"], sd[p + f"mlp.{n}.bias"]
    return out
