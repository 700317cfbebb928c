"""Tensor-level wrappers over the C ABI: torch provides device memory and the stream, libfreepose_hip.so does
the work.  Every function raises if the library is missing or a call fails (no CPU fallback)."""
from __future__ import annotations

import ctypes as C
import functools
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr

_ctx = {}


def context(device: Optional[int] = None):
    """One fp_ctx per (process, device)."""
    if device is None:
        device = torch.cuda.current_device()
    h = _ctx.get(device)
    if h is None:
        lib = _lib.load()
        out = C.c_void_p()
        check(lib.fp_ctx_create(int(device), C.byref(out)), "fp_ctx_create")
        h = _ctx[device] = out
    return h


CTX_OPTIONS = ("ln_fused", "raster_tiled", "gemm_row_split")
BANK_QUERIES_PER_PASS = 4      # bank_scan_kernel: up to 4 queries share one pass over the bank (retrieval.hip fp_bank_scan)


def set_option(name: str, value: int, device: Optional[int] = None):
    """Run-time option of this process's context on `device` (fp_ctx_set_option): 'ln_fused', 'raster_tiled', 'gemm_row_split';
    value < 0 restores the default.  Any other name is a measurement toggle of the LAB build ('gemm_variant', 'gemm_dbg',
    'attn_slots', 'attn_variant', 'topk_select': tools/ call _lib.use_lab() first) and raises on the product library."""
    lib = _lib.load()
    if name in CTX_OPTIONS:
        check(lib.fp_ctx_set_option(context(device), name.encode(), int(value)), "fp_ctx_set_option")
    elif hasattr(lib, "fp_lab_set_option"):
        check(lib.fp_lab_set_option(name.encode(), int(value)), "fp_lab_set_option")
    else:
        raise RuntimeError(f"option '{name}' exists only in the lab build (python -m freepose_amd.build --lab; _lib.use_lab())")


def _dev(t: torch.Tensor, dtype=None) -> torch.Tensor:
    if not t.is_cuda:
        t = t.to("cuda", non_blocking=False)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
VIT_ARCHS = {
    # name: (dim, depth, heads, n_reg).  Only the *_reg hub models: they interpolate the position embedding with
    # antialias=True, offset 0 (what fp_vit_forward implements); the non-reg hub entries use antialias=False and
    # interpolate_offset=0.1 and are not used by the reference (dino.py:8, tracking_refiner.py:23).
    "dinov2_vits14_reg": (384, 12, 6, 4), "dinov2_vitb14_reg": (768, 12, 12, 4), "dinov2_vitl14_reg": (1024, 24, 16, 4),
}
# "patch_normalized": patch features with every row F.normalize()d by the final-norm kernel itself (== l2_normalize(patch features), bit
# for bit) — what the estimators score hypotheses with (pose_estimator.py:85-88): no separate normalisation pass over 1.6 GB
FEATURE_TYPES = {"cls": 0, "reg": 1, "patch": 2, "patch_normalized": 3}


class ViT:
    """Device-resident DINOv2 ViT (hub state-dict layout) driving fp_vit_forward."""

    def __init__(self, model_name: str = "dinov2_vitl14_reg", state_dict: Optional[dict] = None, seed: int = 0,
                 device: Optional[int] = None):
        if model_name not in VIT_ARCHS:
            raise ValueError(f"unknown DINOv2 model {model_name}")
        dim, depth, heads, n_reg = VIT_ARCHS[model_name]
        self.dim, self.depth, self.heads, self.n_reg, self.patch, self.pos_grid = dim, depth, heads, n_reg, 14, 37
        self.lib = _lib.load()
        self.ctx = context(device)
        arch = _lib.VitArch(dim, depth, heads, 4 * dim, 14, n_reg, 37, 1e-6)
        h = C.c_void_p()
        check(self.lib.fp_vit_create(self.ctx, C.byref(arch), C.byref(h)), "fp_vit_create")
        self.handle = h
        self.weights = {}
        if state_dict is None:
            state_dict = random_state_dict(model_name, seed)
        self.load_state_dict(state_dict)

    def load_state_dict(self, sd: dict):
        s = current_stream()
        for name, t in sd.items():
            if name == "mask_token":
                continue
            w = _dev(torch.as_tensor(t), torch.bfloat16)
            self.weights[name] = w  # keep alive: the library stores raw pointers
            check(self.lib.fp_vit_set_weight(self.handle, name.encode(), ptr(w), w.numel(), s), f"set_weight({name})")
        torch.cuda.synchronize()

    def forward(self, images: torch.Tensor, layer: int = 22, feature_type: str = "cls",
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`out`: optional preallocated contiguous bf16 tensor of the result shape (e.g. a slice along dim 0 of a larger
        buffer, so chunked calls need no concatenation)"""
        x = _dev(images, torch.bfloat16)
        B, Cc, H, W = x.shape
        assert Cc == 3
        P = (H // self.patch) * (W // self.patch)
        ft = FEATURE_TYPES[feature_type]
        shape = {0: (B, self.dim), 1: (B, self.n_reg, self.dim), 2: (B, P, self.dim), 3: (B, P, self.dim)}[ft]
        if out is None:
            out = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
        else:
            assert tuple(out.shape) == shape and out.dtype == torch.bfloat16 and out.is_contiguous() and out.device == x.device
        if B > 0:
            check(self.lib.fp_vit_forward(self.handle, ptr(x), B, H, W, int(layer), ft, ptr(out), current_stream()),
                  "fp_vit_forward")
        return out

    __call__ = forward

    def flops(self, B, H, W, layer=22) -> float:
        return float(self.lib.fp_vit_flops(self.handle, B, H, W, layer))

    def profile(self, enable: bool):
        check(self.lib.fp_vit_profile(self.handle, int(enable)))

    def profile_read(self):
        g, a, o, f = C.c_float(), C.c_float(), C.c_float(), C.c_double()
        check(self.lib.fp_vit_profile_read(self.handle, C.byref(g), C.byref(a), C.byref(o), C.byref(f)))
        return {"ms_gemm": g.value, "ms_attn": a.value, "ms_other": o.value, "gemm_flops": f.value,
                "gemm_launches": int(self.lib.fp_vit_profile_gemm_launches(self.handle))}

    def __del__(self):
        try:
            self.lib.fp_vit_destroy(self.handle)
        except Exception:
            pass


def random_state_dict(model_name: str, seed: int = 0) -> dict:
    """Seeded random-init weights with the hub DINOv2 state-dict names/shapes (no checkpoints offline).
    trunc-normal(0.02) linears, LayerScale gamma 1.0, LayerNorm (1, 0) — SURVEY.md §8d C2."""
    dim, depth, heads, n_reg = VIT_ARCHS[model_name]
    g = torch.Generator().manual_seed(seed)

    def tn(*shape, std=0.02):
        return torch.nn.init.trunc_normal_(torch.empty(*shape), std=std, a=-2 * std, b=2 * std, generator=g)

    sd = {"cls_token": tn(1, 1, dim, std=1e-6) + 0.0, "pos_embed": tn(1, 1 + 37 * 37, dim),
          "patch_embed.proj.weight": tn(dim, 3, 14, 14), "patch_embed.proj.bias": tn(dim),
          "norm.weight": torch.ones(dim), "norm.bias": torch.zeros(dim)}
    sd["cls_token"] = tn(1, 1, dim)
    if n_reg:
        sd["register_tokens"] = tn(1, n_reg, dim)
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = torch.ones(dim) + tn(dim, std=0.05)
        sd[p + "norm1.bias"] = tn(dim)
        sd[p + "attn.qkv.weight"] = tn(3 * dim, dim)
        sd[p + "attn.qkv.bias"] = tn(3 * dim)
        sd[p + "attn.proj.weight"] = tn(dim, dim)
        sd[p + "attn.proj.bias"] = tn(dim)
        sd[p + "ls1.gamma"] = torch.ones(dim)
        sd[p + "norm2.weight"] = torch.ones(dim) + tn(dim, std=0.05)
        sd[p + "norm2.bias"] = tn(dim)
        sd[p + "mlp.fc1.weight"] = tn(4 * dim, dim)
        sd[p + "mlp.fc1.bias"] = tn(4 * dim)
        sd[p + "mlp.fc2.weight"] = tn(dim, 4 * dim)
        sd[p + "mlp.fc2.bias"] = tn(dim)
        sd[p + "ls2.gamma"] = torch.ones(dim)
    return {k: v.to(torch.bfloat16) for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------
def ffa(feats: torch.Tensor, masks: torch.Tensor, cell: int = 14, normalize: bool = False, out_f32: bool = False):
    """FFA descriptor.  feats bf16 [B,P,D]; masks bool/u8 [B,gh*cell,gw*cell] (or [B,P] with cell=1)."""
    lib = _lib.load()
    f = _dev(feats, torch.bfloat16)
    B, Pn, D = f.shape
    m = _dev(masks)
    m = (m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)).contiguous()   # (a bool tensor IS 0 / 1 bytes: no copy kernel)
    if cell == 1:
        gh, gw = 1, Pn
        m = m.reshape(B, 1, Pn)
    else:
        gh, gw = m.shape[1] // cell, m.shape[2] // cell
        if m.shape[1] != gh * cell or m.shape[2] != gw * cell:
            m = m[:, :gh * cell, :gw * cell].contiguous()
    assert gh * gw == Pn, f"mask grid {gh}x{gw} != {Pn} patches"
    ob = torch.empty((B, D), dtype=torch.bfloat16, device=f.device)
    of = torch.empty((B, D), dtype=torch.float32, device=f.device) if out_f32 else None
    if B:
        check(lib.fp_ffa(context(), ptr(f), ptr(m), B, gh, gw, D, cell, int(normalize), ptr(ob), ptr(of), current_stream()),
              "fp_ffa")
    return of if out_f32 else ob


def l2_normalize(x: torch.Tensor, inplace: bool = False) -> torch.Tensor:
    """F.normalize(x, dim=-1) with the reference's bf16 rounding points; `inplace` overwrites a contiguous bf16 device tensor"""
    lib = _lib.load()
    xb = _dev(x, torch.bfloat16)
    D = xb.shape[-1]
    rows = xb.numel() // D
    y = xb if inplace else torch.empty_like(xb)
    if rows:
        check(lib.fp_l2_normalize(context(), ptr(xb), rows, D, ptr(y), current_stream()), "fp_l2_normalize")
    return y


def bank_prepare(bank_f32: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    b = _dev(bank_f32, torch.float32)
    N, D = b.shape
    out = torch.empty((N, D), dtype=torch.bfloat16, device=b.device)
    check(lib.fp_bank_prepare(context(), ptr(b), N, D, ptr(out), current_stream()), "fp_bank_prepare")
    return out


def bank_topk(bank_bf16: torch.Tensor, queries: torch.Tensor, k: int = 100, idx_offset: int = 0):
    """(scores f32 [Q,k], idx i32 [Q,k]) ordered by (score desc, index asc)."""
    lib = _lib.load()
    b = _dev(bank_bf16, torch.bfloat16)
    q = _dev(queries, torch.bfloat16)
    if q.dim() == 1:
        q = q[None]
    N, D = b.shape
    Q = q.shape[0]
    s = torch.empty((Q, k), dtype=torch.float32, device=b.device)
    i = torch.empty((Q, k), dtype=torch.int32, device=b.device)
    if Q:
        check(lib.fp_bank_topk(context(), ptr(b), N, D, ptr(q), Q, k, idx_offset, ptr(s), ptr(i), current_stream()),
              "fp_bank_topk")
    return s, i


def topk_merge(cand_scores: torch.Tensor, cand_idx: torch.Tensor, k: int):
    lib = _lib.load()
    cs = _dev(cand_scores, torch.float32)
    ci = _dev(cand_idx, torch.int32)
    Q, Cn = cs.shape
    s = torch.empty((Q, k), dtype=torch.float32, device=cs.device)
    i = torch.empty((Q, k), dtype=torch.int32, device=cs.device)
    if Q:
        check(lib.fp_topk_merge(context(), ptr(cs), ptr(ci), Q, Cn, k, ptr(s), ptr(i), current_stream()), "fp_topk_merge")
    return s, i


def rerank_views(views_bf16: torch.Tensor, offsets: torch.Tensor, cand_idx: torch.Tensor, queries: torch.Tensor, k: int) -> torch.Tensor:
    """per-view fine re-rank scores f32 [Q,C] (numpy-mean of the top-k per-view scores of each candidate mesh)"""
    lib = _lib.load()
    v = _dev(views_bf16, torch.bfloat16)
    off = _dev(offsets).to(torch.int32).contiguous()
    cd = _dev(cand_idx).to(torch.int32).contiguous()
    q = _dev(queries, torch.bfloat16).reshape(cd.shape[0], v.shape[1])
    out = torch.empty(cd.shape, dtype=torch.float32, device=v.device)
    if cd.numel():
        check(lib.fp_rerank_views(context(), ptr(v), ptr(off), ptr(cd), ptr(q), cd.shape[0], cd.shape[1], v.shape[1], int(k),
                                  ptr(out), current_stream()), "fp_rerank_views")
    return out


def template_score(tmpl: torch.Tensor, query: torch.Tensor, weights: Optional[torch.Tensor] = None,
                   normalized: bool = False) -> torch.Tensor:
    """tmpl bf16 [T,P,D] raw — or, with normalized=True, already l2_normalize()d rows (the pre-normalised feature store; same
    bits out, one streaming pass) —, query bf16 [P,D] used as given -> scores f32 [T] (bf16-valued unless weighted)."""
    lib = _lib.load()
    t = _dev(tmpl, torch.bfloat16)
    q = _dev(query, torch.bfloat16).reshape(-1, t.shape[-1])
    T, Pn, D = t.shape
    assert q.shape[0] == Pn
    w = _dev(weights, torch.float32) if weights is not None else None
    out = torch.empty((T,), dtype=torch.float32, device=t.device)
    if T:
        fn = lib.fp_template_score_normed if normalized else lib.fp_template_score
        check(fn(context(), ptr(t), ptr(q), ptr(w), T, Pn, D, ptr(out), current_stream()), "fp_template_score")
    return out


def crop_resize_pad(images: torch.Tensor, boxes: torch.Tensor, target: int, bbox_extend: float = 0.0,
                    masks: Optional[torch.Tensor] = None, mask_mode: int = 0, out_bf16: bool = False,
                    u8_float_div: bool = False) -> torch.Tensor:
    """images f32 [n_img,C,H,W] or u8 [n_img,H,W,C]; boxes int [n,4] xyxy.  u8 pixels become
    float(double(x)/255) (renderer.py:121) or, with u8_float_div, float(x)/255.f (utils.py:20)."""
    lib = _lib.load()
    if images.dtype == torch.uint8:
        img = _dev(images)
        n_img, H, W, Cc = img.shape
        src = 2 if u8_float_div else 1
    else:
        img = _dev(images, torch.float32)
        n_img, Cc, H, W = img.shape
        src = 0
    bx = _dev(boxes).to(torch.int32).contiguous()
    n = bx.shape[0]
    m = _dev(masks).to(torch.uint8).contiguous() if masks is not None else None
    if mask_mode == 2:
        Cc_out = 1
    else:
        Cc_out = Cc
    out = torch.empty((n, Cc_out, target, target), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=img.device)
    if n:
        check(lib.fp_crop_resize_pad(context(), ptr(img), src, n_img, Cc_out if mask_mode == 2 else Cc, H, W, ptr(bx), n,
                                     float(bbox_extend), int(target), ptr(m), int(mask_mode), ptr(out), int(out_bf16),
                                     current_stream()), "fp_crop_resize_pad")
    return out


def roi_align(images: torch.Tensor, rois: torch.Tensor, output_size, sampling_ratio: int = 2,
              spatial_scale: float = 1.0) -> torch.Tensor:
    """torchvision.ops.roi_align(images, rois, output_size, spatial_scale, sampling_ratio, aligned=False):
    images f32 [N,C,H,W], rois f32 [n,5] (image index, x1, y1, x2, y2) -> f32 [n,C,ph,pw]"""
    lib = _lib.load()
    img = _dev(images, torch.float32)
    r = _dev(rois, torch.float32).reshape(-1, 5).contiguous()
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    N, Cc, H, W = img.shape
    n = r.shape[0]
    out = torch.empty((n, Cc, ph, pw), dtype=torch.float32, device=img.device)
    if n:
        check(lib.fp_roi_align(context(), ptr(img), N, Cc, H, W, ptr(r), n, int(ph), int(pw), int(sampling_ratio),
                               float(spatial_scale), ptr(out), current_stream()), "fp_roi_align")
    return out


def generate_rotations(n: int) -> np.ndarray:
    lib = _lib.load()
    out = np.empty((n, 3, 3), dtype=np.float64)
    check(lib.fp_generate_rotations(n, ptr(out)), "fp_generate_rotations")
    return out


def geodesic_select(grid_f64: torch.Tensor, R_prev: np.ndarray, thresh_deg: float) -> np.ndarray:
    lib = _lib.load()
    g = _dev(grid_f64, torch.float64)
    G = g.shape[0]
    idx = torch.empty((G,), dtype=torch.int32, device=g.device)
    Rp = np.ascontiguousarray(np.asarray(R_prev, dtype=np.float64)[:3, :3])
    n = C.c_int(0)
    check(lib.fp_geodesic_select(context(), ptr(g), G, ptr(Rp), float(thresh_deg), ptr(idx), C.byref(n), current_stream()),
          "fp_geodesic_select")
    return idx[: n.value].cpu().numpy().astype(np.int64)


class Mesh:
    """device copy of a triangle mesh.  `colors` u8 [V,3] selects vertex colours; `uv` f32 [F,3,2] + `texture` u8 [th,tw,3]
    (+ optional diffuse factor `kd` [3]) select per-fragment texture sampling; neither = white."""

    def __init__(self, vertices: np.ndarray, faces: np.ndarray, colors: Optional[np.ndarray] = None, uv: Optional[np.ndarray] = None,
                 texture: Optional[np.ndarray] = None, kd=None):
        self.lib = _lib.load()
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        f = np.ascontiguousarray(faces, dtype=np.int32)
        h = C.c_void_p()
        if uv is not None and texture is not None:
            t = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 3, 2)
            if t.shape[0] != f.shape[0]:
                raise ValueError(f"uv must be per-corner [F,3,2]; got {t.shape} for {f.shape[0]} faces")
            x = np.ascontiguousarray(np.asarray(texture)[:, :, :3], dtype=np.uint8)
            k = None if kd is None else np.ascontiguousarray(kd, dtype=np.float32).reshape(3)
            check(self.lib.fp_mesh_upload_textured(context(), ptr(v), v.shape[0], ptr(f), f.shape[0], ptr(t), ptr(x), x.shape[0],
                                                   x.shape[1], ptr(k), C.byref(h)), "fp_mesh_upload_textured")
        else:
            c = np.ascontiguousarray(colors[:, :3], dtype=np.uint8) if colors is not None else None
            check(self.lib.fp_mesh_upload(context(), ptr(v), v.shape[0], ptr(f), f.shape[0], ptr(c), C.byref(h)), "fp_mesh_upload")
        self.handle, self.V, self.F = h, v.shape[0], f.shape[0]

    def set_filter(self, mode: int):
        """texture minification: 1 = trilinear mip-maps (default), 0 = bilinear level 0 (csrc/raster.hip header)"""
        check(self.lib.fp_mesh_set_filter(self.handle, int(mode)), "fp_mesh_set_filter")
        return self

    def set_cull(self, mode: int):
        """back-face culling: 0 = both sides (default; the reference's callers), 1 = pyrender's default culling (cull_faces=True)"""
        check(self.lib.fp_mesh_set_cull(self.handle, int(mode)), "fp_mesh_set_cull")
        self.cull = int(mode)
        return self

    def set_shading(self, mode: int):
        """1 = gamma output rule (default), 0 = linear (csrc/raster.hip header)"""
        check(self.lib.fp_mesh_set_shading(self.handle, int(mode)), "fp_mesh_set_shading")
        return self

    def set_ambient(self, ambient: float):
        """scene ambient light factor (2 = MeshRenderer's scenes, 5 = TrackingRefiner's)"""
        check(self.lib.fp_mesh_set_ambient(self.handle, float(ambient)), "fp_mesh_set_ambient")
        return self

    def __del__(self):
        try:
            self.lib.fp_mesh_destroy(self.handle)
        except Exception:
            pass


def rasterize(mesh: Mesh, poses: torch.Tensor, scale: float, fx: float, fy: float, cx: float, cy: float, W: int, H: int):
    """poses [Hn,4,4] object->camera (OpenCV).  Returns (rgb u8 [Hn,H,W,3], depth f32 [Hn,H,W]) on device."""
    lib = _lib.load()
    p = _dev(torch.as_tensor(poses), torch.float32)
    Hn = p.shape[0]
    rgb = torch.empty((Hn, H, W, 3), dtype=torch.uint8, device=p.device)
    depth = torch.empty((Hn, H, W), dtype=torch.float32, device=p.device)
    if Hn:
        check(lib.fp_rasterize(context(), mesh.handle, ptr(p), Hn, float(scale), float(fx), float(fy), float(cx), float(cy),
                               int(W), int(H), ptr(rgb), ptr(depth), current_stream()), "fp_rasterize")
    return rgb, depth


def rasterize_extents(mesh: Mesh, poses: torch.Tensor, scale: float, fx: float, fy: float, cx: float, cy: float, W: int, H: int,
                      want_depth: bool = False):
    """rasterize() with the depth image's consumers fused into the tile epilogue: returns (rgb u8 [Hn,H,W,3], depth f32 [Hn,H,W] or
    None, ext f64 [Hn,8] == depth_extents(depth), boxes i32 [Hn,4] == ext[:, :4]).  Without `want_depth` the depth image is never
    written (the pose hot path needs only the extents)."""
    lib = _lib.load()
    p = _dev(torch.as_tensor(poses), torch.float32)
    Hn = p.shape[0]
    rgb = torch.empty((Hn, H, W, 3), dtype=torch.uint8, device=p.device)
    depth = torch.empty((Hn, H, W), dtype=torch.float32, device=p.device) if want_depth else None
    ext = torch.empty((Hn, 8), dtype=torch.float64, device=p.device)
    boxes = torch.empty((Hn, 4), dtype=torch.int32, device=p.device)
    if Hn:
        check(lib.fp_rasterize_extents(context(), mesh.handle, ptr(p), Hn, float(scale), float(fx), float(fy), float(cx), float(cy),
                                       int(W), int(H), ptr(rgb), ptr(depth), ptr(ext), ptr(boxes), current_stream()), "fp_rasterize_extents")
    return rgb, depth, ext, boxes


def project_vertices(mesh: Mesh, poses: torch.Tensor, scale: float, fx: float, fy: float, cx: float, cy: float):
    """vertex stage of the rasteriser: (xy i32 [Hn,V,2] window coordinates in 1/256 px, zc f32 [Hn,V]) on device"""
    lib = _lib.load()
    p = _dev(torch.as_tensor(poses), torch.float32)
    Hn = p.shape[0]
    xy = torch.zeros((Hn, mesh.V, 2), dtype=torch.int32, device=p.device)
    zc = torch.zeros((Hn, mesh.V), dtype=torch.float32, device=p.device)
    if Hn:
        check(lib.fp_project_vertices(context(), mesh.handle, ptr(p), Hn, float(scale), float(fx), float(fy), float(cx), float(cy),
                                      ptr(xy), ptr(zc), current_stream()), "fp_project_vertices")
    return xy, zc


def depth_extents(depth: torch.Tensor, fx: float, fy: float, cx: float, cy: float) -> torch.Tensor:
    """[Hn,8] = xmin,ymin,xmax,ymax (mask bbox incl. <100 px fallback), dx, dy (m), count, 0"""
    lib = _lib.load()
    d = _dev(depth, torch.float32)
    Hn, H, W = d.shape
    out = torch.empty((Hn, 8), dtype=torch.float64, device=d.device)
    if Hn:
        check(lib.fp_depth_extents(context(), ptr(d), Hn, H, W, float(fx), float(fy), float(cx), float(cy), ptr(out),
                                   current_stream()), "fp_depth_extents")
    return out


# ---- kernel-level ops (unit tests / microbenchmarks) ------------------------------------------------
def gemm(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, epi: int = 0, gamma=None, resid=None,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """epi 0: x w^T + b ; 1: gelu(.) ; 2: resid + gamma*(.)   (all bf16, fp32 accumulate)"""
    lib = _lib.load()
    x, w, bias = _dev(x, torch.bfloat16), _dev(w, torch.bfloat16), _dev(bias, torch.bfloat16)
    M, K = x.shape
    N = w.shape[0]
    g = _dev(gamma, torch.bfloat16) if gamma is not None else None
    r = _dev(resid, torch.bfloat16) if resid is not None else None
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    check(lib.fp_op_gemm(context(), ptr(x), K, ptr(w), K, ptr(out), N, ptr(bias), ptr(g), ptr(r), N, M, N, K, epi, current_stream()),
          "fp_op_gemm")
    return out


def gelu_direct(x: torch.Tensor) -> torch.Tensor:
    """bf16(0.5 x (1 + erf(x / sqrt 2))) elementwise, the direct fp32 expression the fc1 epilogue's GELU table is filled from"""
    x = _dev(x, torch.bfloat16)
    y = torch.empty_like(x)
    check(_lib.load().fp_op_gelu(ptr(x), ptr(y), x.numel(), current_stream()), "fp_op_gelu")
    return y


def gemm_vt(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, npad: int, heads: int,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    x, w, bias = _dev(x, torch.bfloat16), _dev(w, torch.bfloat16), _dev(bias, torch.bfloat16)
    M, K = x.shape
    N = w.shape[0]
    B = M // npad
    vt = out if out is not None else torch.zeros((B, heads, 64, npad), dtype=torch.bfloat16, device=x.device)
    check(lib.fp_op_gemm_vt(context(), ptr(x), K, ptr(w), K, ptr(vt), ptr(bias), M, N, K, npad, heads, current_stream()), "fp_op_gemm_vt")
    return vt


def ln_linear(x: torch.Tensor, g_ln: torch.Tensor, b_ln: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, mode: int = 0,
              npad: int = 0, heads: int = 0, eps: float = 1e-6, n_scaled: int = 0, row_scale: float = 1.0) -> torch.Tensor:
    """LayerNorm folded into the consuming linear layer (fp_op_ln_linear): mode 0 LN(x) w^T + b, 1 gelu(.), 2 transposed V store.
    Output features below n_scaled are multiplied by row_scale inside the fold (the ViT's q rows: ATTN_QSCALE)"""
    lib = _lib.load()
    x, w, bias = _dev(x, torch.bfloat16), _dev(w, torch.bfloat16), _dev(bias, torch.bfloat16)
    g_ln, b_ln = _dev(g_ln, torch.bfloat16), _dev(b_ln, torch.bfloat16)
    M, K = x.shape
    N = w.shape[0]
    if mode == 2:
        out = torch.zeros((M // npad, heads, 64, npad), dtype=torch.bfloat16, device=x.device)
    else:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    check(lib.fp_op_ln_linear(context(), ptr(x), M, K, ptr(g_ln), ptr(b_ln), float(eps), ptr(w), N, ptr(bias), int(mode), int(npad),
                              int(heads), int(n_scaled), float(row_scale), ptr(out), current_stream()), "fp_op_ln_linear")
    return out


def gemm_stats(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, resid: torch.Tensor, eps: float = 1e-6):
    """resid + gamma * (x w^T + b) plus the row statistics from the epilogue's partial sums (fp_op_gemm_stats): returns (out,
    stat f32 [M,3] = (mean, sigma, rstd)) — mean and sigma decoded from the two-piece bf16 splits the consuming GEMM reads"""
    lib = _lib.load()
    x, w, bias = _dev(x, torch.bfloat16), _dev(w, torch.bfloat16), _dev(bias, torch.bfloat16)
    g, r = _dev(gamma, torch.bfloat16), _dev(resid, torch.bfloat16)
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    raw = torch.zeros((M, 6), dtype=torch.float32, device=x.device)
    check(lib.fp_op_gemm_stats(context(), ptr(x), K, ptr(w), K, ptr(out), N, ptr(bias), ptr(g), ptr(r), N, M, N, K, float(eps), ptr(raw),
                               current_stream()), "fp_op_gemm_stats")
    rec = raw[:, :4].contiguous().view(torch.bfloat16).float()          # [M, 8]: sh, sl, sh, -mh, -ml, -mh, 0, 0
    stat = torch.stack([-(rec[:, 3] + rec[:, 4]), rec[:, 0] + rec[:, 1], raw[:, 4]], dim=1)
    return out, stat


ATTN_QSCALE = 1.4426950408889634 / 8.0   # log2(e) / sqrt(64): what q_prescaled=True expects the q columns to carry


def attention(qk: torch.Tensor, vt: torch.Tensor, n_tok: int, out: Optional[torch.Tensor] = None, q_prescaled: bool = False) -> torch.Tensor:
    """qk bf16 [B*npad, 2*H*64], vt bf16 [B,H,64,npad] -> o bf16 [B*npad, H*64].  q_prescaled: the q columns already hold
    q * ATTN_QSCALE (what the ViT's LayerNorm-folded qkv layer produces)"""
    lib = _lib.load()
    qk, vt = _dev(qk, torch.bfloat16), _dev(vt, torch.bfloat16)
    B, H, _, npad = vt.shape
    if out is None:
        out = torch.zeros((B * npad, H * 64), dtype=torch.bfloat16, device=qk.device)
    check(lib.fp_op_attention(ptr(qk), 2 * H * 64, ptr(vt), ptr(out), H * 64, B, H, n_tok, npad, int(bool(q_prescaled)), current_stream()),
          "fp_op_attention")
    return out


def im2col_norm(images: torch.Tensor, ps: int = 14, kp: Optional[int] = None) -> torch.Tensor:
    """bf16 crops [B,3,H,W] in [0,1] -> normalised patch rows [B*(H/ps)*(W/ps), kp] (fp_op_im2col_norm; kp defaults to 3*ps*ps rounded
    up to a multiple of 64, what fp_vit_forward uses)"""
    lib = _lib.load()
    images = _dev(images, torch.bfloat16)
    B, _, H, W = images.shape
    kp = kp or (3 * ps * ps + 63) // 64 * 64
    out = torch.empty((B * (H // ps) * (W // ps), kp), dtype=torch.bfloat16, device=images.device)
    check(lib.fp_op_im2col_norm(ptr(images), ptr(out), B, H, W, ps, kp, current_stream()), "fp_op_im2col_norm")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    lib = _lib.load()
    x, gamma, beta = _dev(x, torch.bfloat16), _dev(gamma, torch.bfloat16), _dev(beta, torch.bfloat16)
    rows, D = x.shape
    y = torch.empty_like(x)
    check(lib.fp_op_layernorm(ptr(x), ptr(y), ptr(gamma), ptr(beta), rows, D, float(eps), current_stream()), "fp_op_layernorm")
    return y


def plan_vit_batches(n: int, n_tok: int, max_batch: int = 192, n_cu: int = 256) -> list:
    return list(_plan_vit_batches(int(n), int(n_tok), int(max_batch), int(n_cu)))


@functools.lru_cache(maxsize=256)
def _plan_vit_batches(n: int, n_tok: int, max_batch: int, n_cu: int) -> tuple:
    """Split n crops into ViT batches whose GEMM grids fill whole rounds of the resident (one workgroup per CU) grid.

    The persistent GEMM walks 256x256 tiles with one workgroup per CU; a batch of b crops has ceil(b*npad/256) tile rows,
    and the narrowest linear layers (N = 1024: 4 tile columns) are the ones whose last partial round costs the most.
    Dynamic programme over batch sizes in [max_batch/2, max_batch + max_batch/8]: minimise the total number of rounds,
    then the number of batches.  Host-side policy only — results do not depend on the split.
    """
    if n <= 0:
        return ()
    npad = (n_tok + 15) // 16 * 16
    lo, hi = max(1, max_batch // 2), max_batch + max(1, max_batch // 8)

    def rounds(b):
        return -(-(-(-b * npad // 256) * 4) // n_cu)

    if n <= hi:
        return (n,)
    INF = (1 << 60, 1 << 60)
    best = [INF] * (n + 1)
    back = [0] * (n + 1)
    best[0] = (0, 0)
    for m in range(1, n + 1):
        for b in range(lo, min(hi, m) + 1):
            prev = best[m - b]
            if prev == INF:
                continue
            cand = (prev[0] + rounds(b), prev[1] + 1)
            if cand < best[m]:
                best[m], back[m] = cand, b
    if best[n] == INF:            # n below 2*lo: one batch or an even split
        return (n,) if n <= hi else (n // 2, n - n // 2)
    out, m = [], n
    while m > 0:
        out.append(back[m])
        m -= back[m]
    return tuple(sorted(out, reverse=True))


class Timer:
    """HIP-event timer on the current stream (bench.py)."""

    def __init__(self):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.fp_timer_create(C.byref(h)))
        self.h = h

    def start(self):
        check(self.lib.fp_timer_start(self.h, current_stream()))

    def stop(self):
        check(self.lib.fp_timer_stop(self.h, current_stream()))

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        check(self.lib.fp_timer_elapsed_ms(self.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            self.lib.fp_timer_destroy(self.h)
        except Exception:
            pass
