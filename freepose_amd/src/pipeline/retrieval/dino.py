"""Drop-in for the reference's `src/pipeline/retrieval/dino.py` (DINOv2FeatureExtractor :7-32).

The reference pulls `facebookresearch/dinov2` through torch.hub (dino.py:10) and runs it with cuBLAS/xformers; here the
same forward (normalise -> tokens -> first `layer` blocks -> final norm -> cls/reg/patch slice) is one call into
libfreepose_hip.so (fp_vit_forward): hand-written gfx950 MFMA GEMMs with fused epilogues (bias, GELU, LayerScale + residual; LayerNorm folded into the
consuming GEMM algebraically where the library says so) and LDS-tiled attention.  Weights use the official hub state-dict layout, so `dinov2_vitl14_reg4_pretrain.pth` loads unchanged.
"""
from __future__ import annotations

import os
import warnings
from pathlib import Path

import torch

from freepose_amd import ops

_CKPT_NAMES = {   # the *_reg hub models only (ops.VIT_ARCHS: the non-reg entries interpolate the position embedding differently)
    "dinov2_vits14_reg": "dinov2_vits14_reg4_pretrain.pth", "dinov2_vitb14_reg": "dinov2_vitb14_reg4_pretrain.pth",
    "dinov2_vitl14_reg": "dinov2_vitl14_reg4_pretrain.pth",
}


def _find_checkpoint(model_name: str):
    """FREEPOSE_DINOV2_WEIGHTS (file or directory), then the torch.hub checkpoint cache."""
    cand = []
    env = os.environ.get("FREEPOSE_DINOV2_WEIGHTS")
    if env:
        p = Path(env)
        cand.append(p / _CKPT_NAMES[model_name] if p.is_dir() else p)
    cand.append(Path(torch.hub.get_dir()) / "checkpoints" / _CKPT_NAMES[model_name])
    for c in cand:
        if c.is_file():
            return c
    return None


class DINOv2FeatureExtractor(torch.nn.Module):
    """Same surface as the reference nn.Module (dino.py:7-32): construct, `.to(...)`, `.eval()`, call with
    (images, layer=22, feature_type='cls'|'reg'|'patch').  The weights live in the HIP library's device tables (ops.ViT), not
    in nn.Parameters, so `.to(device, dtype)` is a no-op by construction: the model is bf16 on the GPU, like the reference after
    `.to('cuda', dtype=torch.bfloat16)` (pose_estimator.py:21)."""

    def __init__(self, model_name: str = "dinov2_vitl14_reg", state_dict: dict | None = None, seed: int | None = None,
                 allow_random_weights: bool | None = None):
        """Weights: `state_dict` if given, else the hub checkpoint (FREEPOSE_DINOV2_WEIGHTS file/dir, then torch.hub's
        checkpoint cache — what dino.py:10's torch.hub.load would have downloaded).  A missing checkpoint is an ERROR
        (fail closed) unless random weights were asked for: an explicit `seed`, `allow_random_weights=True`
        (the CLIs' --allow_random_weights) or FREEPOSE_ALLOW_RANDOM_WEIGHTS=1 — benchmarks and tests only."""
        super().__init__()
        self.model_name = model_name
        self.checkpoint = None
        if state_dict is None:
            ckpt = _find_checkpoint(model_name)
            if ckpt is not None:
                state_dict = torch.load(ckpt, map_location="cpu")
                if isinstance(state_dict, dict) and "model" in state_dict and "pos_embed" not in state_dict:
                    state_dict = state_dict["model"]
                self.checkpoint = ckpt
            else:
                if allow_random_weights is None:
                    allow_random_weights = seed is not None or os.environ.get("FREEPOSE_ALLOW_RANDOM_WEIGHTS", "0") == "1"
                if not allow_random_weights:
                    raise FileNotFoundError(
                        f"DINOv2 checkpoint {_CKPT_NAMES[model_name]} not found: set FREEPOSE_DINOV2_WEIGHTS to the file or its "
                        f"directory (looked in {os.environ.get('FREEPOSE_DINOV2_WEIGHTS') or '<unset>'} and "
                        f"{Path(torch.hub.get_dir()) / 'checkpoints'}).  Pass --allow_random_weights / allow_random_weights=True "
                        "to run on seeded random-init weights of the same architecture (benchmarks and tests only).")
                warnings.warn(
                    f"no {_CKPT_NAMES[model_name]} found (set FREEPOSE_DINOV2_WEIGHTS); using seeded random-init weights "
                    "with the DINOv2 shapes — features are NOT meaningful for real images", RuntimeWarning)
        self.model = ops.ViT(model_name, state_dict, seed=0 if seed is None else seed)
        self.num_register_tokens = self.model.n_reg

    def forward(self, images, layer=22, feature_type="cls"):
        with torch.inference_mode():
            return self.model.forward(images, layer=layer, feature_type=feature_type)

    def forward_batched(self, images, layer=22, feature_type="patch", batch_size=128):
        """features of MANY crops (the 600 templates of a mesh) in ViT calls of about `batch_size` crops — the reference's loops slice
        `[i:i + batch_size]` (pose_estimator.py:29-33, extract_retrieval_features.py:49-52).  `batch_size` stays the memory bound (never
        exceeded by more than an eighth); within it the split is chosen so that the GEMM grids run whole rounds of the 256-CU grid
        (ops.plan_vit_batches: 600 crops @420^2 at 128 -> 143/143/143/107/64 instead of 4 x 128 + 88: 34 instead of 37 rounds in the
        N = 1024 layers), and each call writes straight into its slice of one output tensor.  Same features: a crop's features do not
        depend on its batch."""
        from freepose_amd import ops
        n = len(images)
        if n == 0:
            return self.forward(images, layer=layer, feature_type=feature_type)
        m = self.model
        H, W = images.shape[-2:]
        P = (H // m.patch) * (W // m.patch)
        shape = {"cls": (n, m.dim), "reg": (n, m.n_reg, m.dim), "patch": (n, P, m.dim)}[feature_type]
        with torch.inference_mode():
            out = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
            at = 0
            for b in ops.plan_vit_batches(n, P + 1 + m.n_reg, int(batch_size)):
                m.forward(images[at:at + b], layer=layer, feature_type=feature_type, out=out[at:at + b])
                at += b
        return out
