"""ctypes wrapper of oracle/libfp_oracle.so (built from fp_oracle.c by freepose_amd.build.build_oracle()).

TEST INFRASTRUCTURE ONLY — the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under freepose_amd/ (the product) may import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "libfp_oracle.so"
        src = _HERE / "fp_oracle.c"
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o",
                            str(so), str(src), "-lm"], check=True)
        _LIB = C.CDLL(str(so))
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def to_bf16_bits(x) -> np.ndarray:
    """float array -> uint16 bf16 bit patterns (round to nearest even)."""
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint16)


def from_bf16_bits(b) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def torch_to_bits(t) -> np.ndarray:
    import torch
    return t.detach().to("cpu").contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def bits_to_torch(b):
    import torch
    return torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint16).view(np.int16).copy()).view(torch.bfloat16)


def l2norm_rows(x_bits: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x_bits, dtype=np.uint16)
    D = x.shape[-1]
    y = np.empty_like(x)
    lib().fpo_l2norm_rows(_p(x), _p(y), C.c_int(x.size // D), C.c_int(D))
    return y


def bank_prepare(bank_f32: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(bank_f32, dtype=np.float32)
    out = np.empty(b.shape, dtype=np.uint16)
    lib().fpo_bank_prepare(_p(b), _p(out), C.c_int(b.shape[0]), C.c_int(b.shape[1]))
    return out


def bank_scores(bank_bits: np.ndarray, q_bits: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(bank_bits, dtype=np.uint16)
    q = np.ascontiguousarray(q_bits, dtype=np.uint16)
    s = np.empty(b.shape[0], dtype=np.float32)
    lib().fpo_bank_scores(_p(b), _p(q), _p(s), C.c_int(b.shape[0]), C.c_int(b.shape[1]))
    return s


def bank_topk(bank_bits: np.ndarray, q_bits: np.ndarray, k: int, idx_offset: int = 0):
    b = np.ascontiguousarray(bank_bits, dtype=np.uint16)
    q = np.ascontiguousarray(q_bits, dtype=np.uint16).reshape(-1, b.shape[1])
    Q = q.shape[0]
    s = np.empty((Q, k), dtype=np.float32)
    i = np.empty((Q, k), dtype=np.int32)
    lib().fpo_bank_topk(_p(b), C.c_int(b.shape[0]), C.c_int(b.shape[1]), _p(q), C.c_int(Q), C.c_int(k), C.c_int(idx_offset),
                        _p(s), _p(i))
    return s, i


def topk_merge(cs: np.ndarray, ci: np.ndarray, k: int):
    cs = np.ascontiguousarray(cs, dtype=np.float32)
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    Q, Cn = cs.shape
    s = np.empty((Q, k), dtype=np.float32)
    i = np.empty((Q, k), dtype=np.int32)
    for q in range(Q):
        lib().fpo_topk_merge(_p(cs[q]), _p(ci[q]), C.c_int(Cn), C.c_int(k), _p(s[q]), _p(i[q]))
    return s, i


def ffa(feat_bits: np.ndarray, mask_u8: np.ndarray, cell: int):
    f = np.ascontiguousarray(feat_bits, dtype=np.uint16)
    B, P, D = f.shape
    m = np.ascontiguousarray(mask_u8, dtype=np.uint8)
    if cell == 1:
        gh, gw = 1, P
    else:
        gh, gw = m.shape[1] // cell, m.shape[2] // cell
        m = np.ascontiguousarray(m[:, :gh * cell, :gw * cell])
    ob = np.empty((B, D), dtype=np.uint16)
    of = np.empty((B, D), dtype=np.float32)
    lib().fpo_ffa(_p(f), _p(m), C.c_int(B), C.c_int(gh), C.c_int(gw), C.c_int(D), C.c_int(cell), _p(ob), _p(of))
    return ob, of


def template_score(tmpl_bits: np.ndarray, q_bits: np.ndarray, weights: np.ndarray | None = None) -> np.ndarray:
    t = np.ascontiguousarray(tmpl_bits, dtype=np.uint16)
    T, P, D = t.shape
    q = np.ascontiguousarray(q_bits, dtype=np.uint16).reshape(P, D)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    s = np.empty(T, dtype=np.float32)
    lib().fpo_template_score(_p(t), _p(q), _p(w), C.c_int(T), C.c_int(P), C.c_int(D), _p(s))
    return s


def rerank_views(view_bits: np.ndarray, offsets: np.ndarray, cand: np.ndarray, q_bits: np.ndarray, k: int) -> np.ndarray:
    v = np.ascontiguousarray(view_bits, dtype=np.uint16)
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    cd = np.ascontiguousarray(cand, dtype=np.int32)
    q = np.ascontiguousarray(q_bits, dtype=np.uint16).reshape(cd.shape[0], v.shape[1])
    out = np.empty(cd.shape, dtype=np.float32)
    lib().fpo_rerank_views(_p(v), _p(off), _p(cd), _p(q), C.c_int(cd.shape[0]), C.c_int(cd.shape[1]), C.c_int(v.shape[1]),
                           C.c_int(k), _p(out))
    return out


def crop_resize_pad(images: np.ndarray, boxes: np.ndarray, target: int, ext: float = 0.0, masks=None, mask_mode: int = 0,
                    u8_float_div: bool = False):
    if images.dtype == np.uint8:
        img = np.ascontiguousarray(images)
        n_img, H, W, Cc = img.shape
        src = 2 if u8_float_div else 1
    else:
        img = np.ascontiguousarray(images, dtype=np.float32)
        n_img, Cc, H, W = img.shape
        src = 0
    bx = np.ascontiguousarray(boxes, dtype=np.int32)
    n = bx.shape[0]
    m = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    Co = 1 if mask_mode == 2 else Cc
    out = np.empty((n, Co, target, target), dtype=np.float32)
    rc = lib().fpo_crop_resize_pad(_p(img), C.c_int(src), C.c_int(n_img), C.c_int(Co), C.c_int(H), C.c_int(W), _p(bx),
                                   C.c_int(n), C.c_float(ext), C.c_int(target), _p(m), C.c_int(mask_mode), _p(out))
    if rc:
        raise ValueError(f"box {rc - 1} does not resize to {target}")
    return out


def generate_rotations(n: int) -> np.ndarray:
    out = np.empty((n, 3, 3), dtype=np.float64)
    lib().fpo_generate_rotations(C.c_int(n), _p(out))
    return out


def geodesic_select(grid: np.ndarray, R_prev: np.ndarray, thresh_deg: float) -> np.ndarray:
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 9)
    Rp = np.ascontiguousarray(np.asarray(R_prev, dtype=np.float64)[:3, :3])
    idx = np.empty(g.shape[0], dtype=np.int32)
    lib().fpo_geodesic_select.restype = C.c_int
    n = lib().fpo_geodesic_select(_p(g), C.c_int(g.shape[0]), _p(Rp), C.c_double(thresh_deg), _p(idx))
    return idx[:n].astype(np.int64)


def depth_extents(depth: np.ndarray, fx, fy, cx, cy) -> np.ndarray:
    d = np.ascontiguousarray(depth, dtype=np.float32)
    Hn, H, W = d.shape
    out = np.empty((Hn, 8), dtype=np.float64)
    # the C ABI carries the intrinsics as float32 (include/freepose_hip.h); widen exactly those values
    fx, fy, cx, cy = (float(np.float32(v)) for v in (fx, fy, cx, cy))
    lib().fpo_depth_extents(_p(d), C.c_int(Hn), C.c_int(H), C.c_int(W), C.c_double(fx), C.c_double(fy), C.c_double(cx),
                            C.c_double(cy), _p(out))
    return out


def roi_align(images: np.ndarray, rois: np.ndarray, ph: int, pw: int, sampling_ratio: int = 2, spatial_scale: float = 1.0) -> np.ndarray:
    """torchvision.ops.roi_align(..., aligned=False) restated (parity unpinned: torchvision is absent here)"""
    img = np.ascontiguousarray(images, dtype=np.float32)
    r = np.ascontiguousarray(rois, dtype=np.float32).reshape(-1, 5)
    N, Cc, H, W = img.shape
    out = np.empty((r.shape[0], Cc, ph, pw), dtype=np.float32)
    lib().fpo_roi_align(_p(img), C.c_int(N), C.c_int(Cc), C.c_int(H), C.c_int(W), _p(r), C.c_int(r.shape[0]), C.c_int(ph),
                        C.c_int(pw), C.c_int(sampling_ratio), C.c_float(spatial_scale), _p(out))
    return out


def rasterize(verts, faces, colors, poses, scale, fx, fy, cx, cy, W, H, ambient: float = 2.0, shade: int = 1, uv=None, texture=None,
              kd=None, filter: int = 1, cull: int = 0):
    """colors u8 [V,3] | None; uv f32 [F,3,2] + texture u8 [th,tw,3] select the textured path; shade 1 = gamma rule, 0 = linear;
    filter 1 = trilinear mip-maps (default), 0 = bilinear level 0"""
    v = np.ascontiguousarray(verts, dtype=np.float32)
    f = np.ascontiguousarray(faces, dtype=np.int32)
    c = None if colors is None else np.ascontiguousarray(colors[:, :3], dtype=np.uint8)
    p = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
    t = None if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 6)
    x = None if texture is None else np.ascontiguousarray(texture[:, :, :3], dtype=np.uint8)
    k = None if kd is None else np.ascontiguousarray(kd, dtype=np.float32)
    th, tw = (x.shape[0], x.shape[1]) if x is not None else (0, 0)
    assert t is None or t.shape[0] == f.shape[0]
    Hn = p.shape[0]
    rgb = np.empty((Hn, H, W, 3), dtype=np.uint8)
    depth = np.empty((Hn, H, W), dtype=np.float32)
    lib().fpo_rasterize_tex(_p(v), C.c_int(v.shape[0]), _p(f), C.c_int(f.shape[0]), _p(c), _p(t), _p(x), C.c_int(th), C.c_int(tw),
                            _p(k), _p(p), C.c_int(Hn), C.c_float(scale), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                            C.c_int(W), C.c_int(H), _p(rgb), _p(depth), C.c_float(ambient), C.c_int(shade), C.c_int(filter), C.c_int(cull))
    return rgb, depth


def project_vertices(verts, poses, scale, fx, fy, cx, cy):
    """vertex stage of the rasteriser: (xy i32 [Hn,V,2] in 1/256 px, zc f32 [Hn,V])"""
    v = np.ascontiguousarray(verts, dtype=np.float32)
    p = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
    Hn, V = p.shape[0], v.shape[0]
    xy = np.empty((Hn, V, 2), dtype=np.int32)
    zc = np.empty((Hn, V), dtype=np.float32)
    lib().fpo_project_vertices(_p(v), C.c_int(V), _p(p), C.c_int(Hn), C.c_float(scale), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                               C.c_float(cy), _p(xy), _p(zc))
    return xy, zc


def shade_tables():
    dec = np.empty(256, np.float32)
    thr = np.empty(256, np.float32)
    lib().fpo_shade_tables(_p(dec), _p(thr))
    return dec, thr
