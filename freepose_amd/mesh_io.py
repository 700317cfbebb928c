"""Minimal triangle-mesh container + Wavefront OBJ reader (the reference uses trimesh.load(..., force='mesh'),
scripts/dino_inference_video.py:93-101; trimesh is not a dependency here).  Per-vertex colours come from `v x y z r g b`
records, or from the diffuse texture sampled (nearest texel) at each vertex's UV, else white."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import numpy as np


@dataclass
class TriMesh:
    vertices: np.ndarray                 # [V,3] float
    faces: np.ndarray                    # [F,3] int
    vertex_colors: Optional[np.ndarray] = None   # [V,3] uint8

    def apply_scale(self, s: float):     # trimesh-compatible (reference mutates the mesh, online_pose_estimator.py:60,64)
        self.vertices = self.vertices * s
        return self

    def copy(self):
        return TriMesh(self.vertices.copy(), self.faces.copy(), None if self.vertex_colors is None else self.vertex_colors.copy())


def mesh_arrays(mesh):
    """(vertices, faces, colors|None) from a TriMesh or any trimesh-like object."""
    v = np.asarray(mesh.vertices, dtype=np.float32)
    f = np.asarray(mesh.faces, dtype=np.int32)
    c = getattr(mesh, "vertex_colors", None)
    if c is None:
        vis = getattr(mesh, "visual", None)
        c = getattr(vis, "vertex_colors", None) if vis is not None else None
    if c is not None:
        c = np.asarray(c)[:, :3].astype(np.uint8)
        if len(c) != len(v):
            c = None
    return v, f, c


def load_obj(path) -> TriMesh:
    path = Path(path)
    vs, vcol, vts, faces, face_vt = [], [], [], [], []
    mtllib = None
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
        elif t[0] == "mtllib":
            mtllib = " ".join(t[1:])
    V = np.asarray(vs, dtype=np.float32).reshape(-1, 3)
    F = np.asarray(faces, dtype=np.int32).reshape(-1, 3)
    colors = None
    if len(vcol) == len(vs) and vs:
        c = np.asarray(vcol, dtype=np.float32)
        colors = np.clip(c * (255.0 if c.max() <= 1.0 else 1.0) + 0.5, 0, 255).astype(np.uint8)
    elif mtllib and vts:
        tex = _diffuse_texture(path.parent / mtllib)
        if tex is not None:
            uv = np.zeros((len(V), 2), np.float32)
            fv, ft = F.reshape(-1), np.asarray(face_vt, dtype=np.int64).reshape(-1)
            ok = ft >= 0
            uv[fv[ok]] = np.asarray(vts, np.float32)[ft[ok]]
            h, w = tex.shape[:2]
            x = np.clip((uv[:, 0] % 1.0) * w, 0, w - 1).astype(int)
            y = np.clip((1.0 - (uv[:, 1] % 1.0)) * h, 0, h - 1).astype(int)
            colors = tex[y, x, :3].astype(np.uint8)
    return TriMesh(V, F, colors)


def _diffuse_texture(mtl_path: Path):
    if not mtl_path.is_file():
        return None
    for line in mtl_path.read_text(errors="ignore").splitlines():
        t = line.split()
        if t and t[0] == "map_Kd":
            p = mtl_path.parent / t[-1]
            if p.is_file():
                try:
                    from PIL import Image
                    return np.asarray(Image.open(p).convert("RGB"))
                except Exception:
                    return None
    return None
