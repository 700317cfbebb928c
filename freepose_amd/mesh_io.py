"""Triangle-mesh container + Wavefront OBJ/MTL reader (the reference uses trimesh.load(path, force='mesh'),
scripts/dino_inference_video.py:93-101, scripts/render_templates.py:58-66; trimesh is not a dependency here).

Appearance, in the order the rasteriser uses it:
  * per-corner texture coordinates `uv` [F,3,2] + diffuse `texture` [h,w,3] (+ material factor `kd`) -> per-fragment texture
    sampling (what pyrender.Mesh.from_trimesh renders for a TextureVisuals mesh, renderer.py:43-45);
  * per-vertex colours (`v x y z r g b` records, trimesh ColorVisuals);
  * white.
Several materials (`usemtl` groups) are packed into one atlas, like trimesh's concatenation under force='mesh': textures are
stacked vertically (each scaled to the widest one with nearest sampling), untextured materials become solid Kd patches, and a
face's coordinates are moved into its material's band (REPEAT addressing inside a band is lost for faces that span more than
one texture period — the same restriction as trimesh's packer).
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import numpy as np


@dataclass
class TriMesh:
    vertices: np.ndarray                          # [V,3] float
    faces: np.ndarray                             # [F,3] int
    vertex_colors: Optional[np.ndarray] = None    # [V,3] uint8
    uv: Optional[np.ndarray] = None               # [F,3,2] float32, OBJ convention (v up)
    texture: Optional[np.ndarray] = None          # [h,w,3] uint8, rows top to bottom
    kd: Optional[np.ndarray] = None               # [3] float32 diffuse factor applied to the texel

    def apply_scale(self, s: float):     # trimesh-compatible (reference mutates the mesh, online_pose_estimator.py:60,64)
        self.vertices = self.vertices * s
        return self

    def copy(self):
        c = lambda a: None if a is None else np.array(a, copy=True)   # noqa: E731
        return TriMesh(c(self.vertices), c(self.faces), c(self.vertex_colors), c(self.uv), c(self.texture), c(self.kd))


def mesh_appearance(mesh) -> dict:
    """keyword arguments for ops.Mesh from a TriMesh or a trimesh-like object: {'colors'} | {'uv','texture','kd'} | {}"""
    faces = np.asarray(mesh.faces)
    n_v = len(np.asarray(mesh.vertices))
    uv, tex, kd = getattr(mesh, "uv", None), getattr(mesh, "texture", None), getattr(mesh, "kd", None)
    if uv is not None and tex is not None:
        return {"uv": np.asarray(uv, np.float32).reshape(-1, 3, 2), "texture": np.asarray(tex)[:, :, :3].astype(np.uint8), "kd": kd}
    vis = getattr(mesh, "visual", None)
    if vis is not None and getattr(vis, "uv", None) is not None:       # trimesh TextureVisuals: per-vertex uv + material image
        mat = getattr(vis, "material", None)
        img = getattr(mat, "image", None)
        if img is None:
            img = getattr(mat, "baseColorTexture", None)
        if img is not None:
            tex = np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)[:, :, :3].astype(np.uint8)
            vuv = np.asarray(vis.uv, np.float32)
            if len(vuv) == n_v:
                fac = getattr(mat, "diffuse", None)
                if fac is None:
                    fac = getattr(mat, "baseColorFactor", None)
                kd = None
                if fac is not None:
                    fac = np.asarray(fac, np.float32).reshape(-1)[:3]
                    kd = fac / 255.0 if fac.max() > 1.0 else fac
                return {"uv": vuv[faces.reshape(-1)].reshape(-1, 3, 2), "texture": tex, "kd": kd}
        if hasattr(vis, "to_color"):                                    # textured visual without a usable image
            vis = vis.to_color()
    c = getattr(mesh, "vertex_colors", None)
    if c is None and vis is not None:
        c = getattr(vis, "vertex_colors", None)
    if c is not None:
        c = np.asarray(c)
        if c.ndim == 2 and len(c) == n_v:
            return {"colors": c[:, :3].astype(np.uint8)}
    return {}


def mesh_arrays(mesh):
    """(vertices f32 [V,3], faces i32 [F,3], appearance dict) from a TriMesh or any trimesh-like object."""
    v = np.asarray(mesh.vertices, dtype=np.float32)
    f = np.asarray(mesh.faces, dtype=np.int32)
    return v, f, mesh_appearance(mesh)


def device_mesh(mesh):
    """upload a TriMesh / trimesh-like object: ops.Mesh with its textured, vertex-coloured or plain appearance"""
    from freepose_amd import ops
    v, f, app = mesh_arrays(mesh)
    return ops.Mesh(v, f, **app)


def mesh_signature(mesh):
    """cheap content check for device-mesh caches: meshes are mutated in place by the pipeline (apply_scale,
    online_pose_estimator.py:60,64) and ids are recycled, so identity alone is not a key"""
    v = np.asarray(mesh.vertices)
    f = np.asarray(mesh.faces)
    return (v.shape, f.shape, float(np.abs(v).sum()), float(v[0].sum()) if len(v) else 0.0, int(f[-1].sum()) if len(f) else 0)


# ---- OBJ / MTL -------------------------------------------------------------------------------------------------------
def _read_mtl(mtl_path: Path) -> dict:
    """material name -> {'kd': [3] float, 'map': Path | None}"""
    mats, cur = {}, None
    if not mtl_path.is_file():
        return mats
    for line in mtl_path.read_text(errors="ignore").splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "newmtl":
            cur = " ".join(t[1:])
            mats[cur] = {"kd": None, "map": None}
        elif cur is not None and t[0] == "Kd" and len(t) >= 4:
            mats[cur]["kd"] = np.array([float(x) for x in t[1:4]], np.float32)
        elif cur is not None and t[0] == "map_Kd":
            p = mtl_path.parent / t[-1]
            mats[cur]["map"] = p if p.is_file() else None
    return mats


def _load_image(p: Path):
    try:
        from PIL import Image
        im = Image.open(p)
        if im.mode in ("LA", "1", "P", "L", "RGBA", "I;16"):   # fix_mesh_texture's conversions (render_templates.py:13-25) and more
            im = im.convert("RGB")
        return np.asarray(im.convert("RGB"))
    except Exception:
        return None


def load_obj(path) -> TriMesh:
    path = Path(path)
    vs, vcol, vts, faces, face_vt, face_mat = [], [], [], [], [], []
    mtllib, cur_mat = None, None
    for line in path.read_text(errors="ignore").splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            vs.append([float(x) for x in t[1:4]])
            if len(t) >= 7:
                vcol.append([float(x) for x in t[4:7]])
        elif t[0] == "vt":
            vts.append([float(t[1]), float(t[2]) if len(t) > 2 else 0.0])
        elif t[0] == "f":
            idx = [p.split("/") for p in t[1:]]
            vi = [int(p[0]) for p in idx]
            ti = [int(p[1]) if len(p) > 1 and p[1] else 0 for p in idx]
            vi = [i - 1 if i > 0 else len(vs) + i for i in vi]
            ti = [i - 1 if i > 0 else (len(vts) + i if i < 0 else -1) for i in ti]
            for k in range(1, len(vi) - 1):     # fan triangulation
                faces.append([vi[0], vi[k], vi[k + 1]])
                face_vt.append([ti[0], ti[k], ti[k + 1]])
                face_mat.append(cur_mat)
        elif t[0] == "mtllib":
            mtllib = " ".join(t[1:])
        elif t[0] == "usemtl":
            cur_mat = " ".join(t[1:])
    V = np.asarray(vs, dtype=np.float32).reshape(-1, 3)
    F = np.asarray(faces, dtype=np.int32).reshape(-1, 3)
    if len(vcol) == len(vs) and vs:
        c = np.asarray(vcol, dtype=np.float32)
        colors = np.clip(c * (255.0 if c.max() <= 1.0 else 1.0) + 0.5, 0, 255).astype(np.uint8)
        return TriMesh(V, F, colors)
    mats = _read_mtl(path.parent / mtllib) if mtllib else {}
    if not mats or not len(F):
        return TriMesh(V, F, None)
    return _with_materials(V, F, np.asarray(vts, np.float32).reshape(-1, 2), np.asarray(face_vt, np.int64).reshape(-1, 3),
                           face_mat, mats)


def _with_materials(V, F, vts, face_vt, face_mat, mats) -> TriMesh:
    used = []
    for m in face_mat:
        if m not in used:
            used.append(m)
    if None in used and len(mats) == 1:         # faces before any usemtl take the only material
        only = next(iter(mats))
        face_mat = [only if m is None else m for m in face_mat]
        used = [only if m is None else m for m in used]
        used = list(dict.fromkeys(used))
    images = {}
    for m in used:
        info = mats.get(m)
        img = _load_image(info["map"]) if info and info["map"] is not None else None
        images[m] = img
    has_uv = len(vts) > 0
    if len(used) == 1 and images[used[0]] is not None and has_uv:
        m = used[0]
        ok = (face_vt >= 0).all(axis=1)
        uv = np.zeros((len(F), 3, 2), np.float32)
        uv[ok] = vts[face_vt[ok]]
        kd = mats[m]["kd"] if m in mats else None
        # a textured material's Kd multiplies the texel (baseColorFactor); exporters write 0 0 0 next to map_Kd when they mean "texture only"
        if kd is not None and float(kd.max()) <= 0.0:
            kd = None
        return TriMesh(V, F, None, uv, images[m], kd)
    if all(images[m] is None for m in used):
        kds = [mats[m]["kd"] if (m in mats and mats[m]["kd"] is not None) else np.ones(3, np.float32) for m in used]
        if len(used) == 1:                       # one untextured material: a uniform colour
            col = np.clip(kds[0] * 255.0 + 0.5, 0, 255).astype(np.uint8)
            return TriMesh(V, F, np.tile(col, (len(V), 1)))
    # ---- atlas: one band per material, stacked vertically ---------------------------------------------------------
    width = max([images[m].shape[1] for m in used if images[m] is not None] + [4])
    bands, tops = [], {}
    y = 0
    for m in used:
        img = images[m]
        if img is None:
            kd = mats[m]["kd"] if (m in mats and mats[m]["kd"] is not None) else np.ones(3, np.float32)
            band = np.tile(np.clip(kd * 255.0 + 0.5, 0, 255).astype(np.uint8), (4, width, 1))
        else:
            h0, w0 = img.shape[:2]
            if w0 != width:                      # nearest resample to the common width (keeps the aspect ratio)
                h1 = max(1, int(round(h0 * width / w0)))
                yy = np.minimum((np.arange(h1) * h0 / h1).astype(int), h0 - 1)
                xx = np.minimum((np.arange(width) * w0 / width).astype(int), w0 - 1)
                img = img[yy][:, xx]
            band = img[:, :, :3]
            kd = mats[m]["kd"] if m in mats else None
            if kd is not None and float(kd.max()) > 0.0 and not np.allclose(kd, 1.0):
                band = np.clip(band.astype(np.float32) * kd + 0.5, 0, 255).astype(np.uint8)
        tops[m] = (y, band.shape[0])
        bands.append(band)
        y += band.shape[0]
    atlas = np.concatenate(bands, axis=0)
    H = atlas.shape[0]
    uv = np.zeros((len(F), 3, 2), np.float32)
    for i in range(len(F)):
        top, hb = tops[face_mat[i]]
        if images[face_mat[i]] is None or not has_uv or (face_vt[i] < 0).any():
            u, v = np.full(3, 0.5, np.float32), np.full(3, 0.5, np.float32)
        else:
            t = vts[face_vt[i]]
            u = t[:, 0] - np.floor(t[:, 0].min())
            v = t[:, 1] - np.floor(t[:, 1].min())
            u, v = np.clip(u, 0.0, 1.0), np.clip(v, 0.0, 1.0)
        # rows of the band, half a texel inside so that bilinear taps stay in the band
        vy = top + 0.5 + (1.0 - v) * (hb - 1.0)
        uv[i, :, 0] = u
        uv[i, :, 1] = 1.0 - vy / H
    return TriMesh(V, F, None, uv, atlas, None)
