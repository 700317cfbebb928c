"""Synthetic end-to-end workspace in the reference's on-disk layout, for the CLI tests and bench.py's video
workload: meshes (one vertex-coloured, one UV-textured OBJ) -> template shards (scripts.render_templates on the HIP rasteriser)
-> video frames / BOP images rendered from known poses -> proposals JSON (RLE masks, xywh boxes, mesh ids, scales).
Test data builder: uses the product's renderer to draw the frames, nothing from oracle/."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch
from PIL import Image

MESH_IDS = ["balla", "cubet"]      # no underscores, like the shipped mesh list (the loader strips them from its index, template.py:41)


def write_meshes(root: Path):
    import bench
    from tests._meshes import checker_gradient_texture, write_textured_obj
    mc = root / "data" / "mesh_cache"
    v, f, c = bench.synthetic_mesh(3, seed=41)
    d = mc / "balla"
    d.mkdir(parents=True, exist_ok=True)
    with open(d / "balla.obj", "w") as fh:
        for p, col in zip(v, c):
            fh.write(f"v {p[0]:.7f} {p[1]:.7f} {p[2]:.7f} {col[0] / 255:.6f} {col[1] / 255:.6f} {col[2] / 255:.6f}\n")
        for t in f:
            fh.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
    write_textured_obj(mc / "cubet", "cubet", checker_gradient_texture(128))
    (root / "data" / "mesh_cache.txt").write_text("\n".join(MESH_IDS) + "\n")
    (root / "data" / "mesh_cache.csv").write_text("model_name\n" + "\n".join(MESH_IDS) + "\n")


def render_shards(root: Path, n_views: int):
    from scripts import render_templates
    return render_templates.run(["--filelist", str(root / "data" / "mesh_cache.txt"), "--mesh_root", str(root / "data" / "mesh_cache"),
                                 "--datasets_root", str(root / "data" / "datasets"), "--shards_folder", "objaverse_shards",
                                 "--n_views", str(n_views)])


def _slow_rotation(R0, step_deg, k):
    from scipy.spatial.transform import Rotation as Rot
    return Rot.from_rotvec(np.deg2rad(step_deg * k) * np.array([0.3, 1.0, 0.2]) / np.linalg.norm([0.3, 1.0, 0.2])).as_matrix() @ R0


def draw_frames(root: Path, n_frames: int, n_views: int, size=(480, 640), scales=(0.10, 0.08), seed=5, step_deg=3.0):
    """-> (frames u8 [F,H,W,3], props: list per frame of per-object dicts, gt poses [F,n_obj,4,4], K); the frames' z-buffers
    (metres, 0 = background) are left in `draw_frames.depths` for write_bop(..., depths=...)"""
    from freepose_amd import ops
    from freepose_amd.mesh_io import device_mesh, load_obj
    from freepose_amd.src.pipeline.retrieval.renderer import grid_poses
    from freepose_amd.src.pipeline.utils import mask_to_rle_pytorch
    H, W = size
    f = float(np.sqrt(H ** 2 + W ** 2))
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]])
    grid = np.array(grid_poses(n_views))
    meshes = [device_mesh(load_obj(root / "data" / "mesh_cache" / m / f"{m}.obj")) for m in MESH_IDS]
    rng = np.random.Generator(np.random.PCG64(seed))
    centres = [(-0.16, -0.02, 0.85), (0.17, 0.05, 0.8)]
    start = [5 % n_views, 17 % n_views]
    frames, props, gts, depths = [], [], [], []
    for fr in range(n_frames):
        img = rng.integers(0, 60, size=(H, W, 3), dtype=np.uint8)
        zbuf = np.full((H, W), np.inf, np.float32)
        per_obj, gt = [], []
        for o, mesh in enumerate(meshes):
            P = np.eye(4)
            P[:3, :3] = _slow_rotation(grid[start[o]][:3, :3], step_deg, fr)
            P[:3, 3] = np.array(centres[o]) + np.array([0.004 * fr, -0.002 * fr, 0.0])
            rgb, depth = ops.rasterize(mesh, torch.from_numpy(P[None].astype(np.float32)), scales[o], f, f, W / 2.0, H / 2.0, W, H)
            rgb, depth = rgb[0].cpu().numpy(), depth[0].cpu().numpy()
            vis = (depth > 0) & (depth < zbuf)
            img[vis] = rgb[vis]
            zbuf[vis] = depth[vis]
            per_obj.append(depth > 0)
            gt.append(P)
        entries = []
        for o, m in enumerate(per_obj):
            ys, xs = np.nonzero(m)
            x0, y0, x1, y1 = int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())
            rle = mask_to_rle_pytorch(torch.from_numpy(m[None]))[0]
            entries.append({"bbox": [x0, y0, x1 - x0, y1 - y0], "segmentation": rle, "mesh": MESH_IDS[o], "score": 0.9 - 0.1 * o,
                            "scene_id": 0, "image_id": fr, "time": 0.01, "scale": float(scales[o])})
        frames.append(img)
        props.append(entries)
        gts.append(gt)
        depths.append(np.where(np.isfinite(zbuf), zbuf, 0.0).astype(np.float32))
    draw_frames.depths = depths
    return np.stack(frames), props, np.array(gts), K


def write_video(root: Path, video: str, frames, props, proposals_name="props.json"):
    vd = root / "data" / "datasets" / "videos" / video
    vd.mkdir(parents=True, exist_ok=True)
    for i, fr in enumerate(frames):
        Image.fromarray(fr, "RGB").save(vd / f"{i:05d}.jpg", quality=97)
    rd = root / "data" / "results" / "videos" / video
    rd.mkdir(parents=True, exist_ok=True)
    (rd / proposals_name).write_text(json.dumps([e for fr in props for e in fr]))
    return rd / proposals_name


def write_bop(root: Path, dataset: str, frames, props, K, scene=48, proposals_name="props.json", depths=None):
    """BOP layout; `depths` (metres) are written as 16-bit PNGs in units of 0.1 mm, the unit the BOPDataset mirror assumes"""
    sd = root / "data" / "datasets" / dataset / "test" / f"{scene:06d}"
    (sd / "rgb").mkdir(parents=True, exist_ok=True)
    if depths is not None:
        (sd / "depth").mkdir(parents=True, exist_ok=True)
    cam = {}
    flat = []
    for i, fr in enumerate(frames):
        Image.fromarray(fr, "RGB").save(sd / "rgb" / f"{i + 1:06d}.png")
        if depths is not None:
            Image.fromarray(np.round(depths[i] * 10000.0).astype(np.uint16)).save(sd / "depth" / f"{i + 1:06d}.png")
        cam[str(i + 1)] = {"cam_K": [float(x) for x in K.reshape(-1)], "depth_scale": 1.0}
        for e in props[i]:
            flat.append(dict(e, scene_id=scene, image_id=i + 1))
    (sd / "scene_camera.json").write_text(json.dumps(cam))
    rd = root / "data" / "results" / dataset
    rd.mkdir(parents=True, exist_ok=True)
    (rd / proposals_name).write_text(json.dumps(flat))
    return rd / proposals_name


def rotation_error_deg(Ra, Rb):
    c = (np.trace(Ra @ Rb.T) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))
