"""TrackingRefiner.n_inliers_per_pose on a 32-frame clip: the pairs one by one (two B = 1 ViT-B/14 forwards @518^2 each, the reference's
loop) against windows of 16 pairs per ViT call (pose_confidences).    python tools/refiner_window_ab.py"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from freepose_amd import ops  # noqa: E402
from freepose_amd.mesh_io import TriMesh  # noqa: E402
from freepose_amd.src.pipeline.estimators.tracking_refiner import TrackingRefiner  # noqa: E402
from freepose_amd.src.pipeline.retrieval.renderer import grid_poses  # noqa: E402

v, f, c = bench.synthetic_mesh(4)
mesh = TriMesh(v * 0.25, f, c)
K = np.array([[600.0, 0, 320.0], [0, 600.0, 240.0], [0, 0, 1]])
poses = np.array(grid_poses(40))
Ts = []
for i in range(32):
    T = poses[i].copy()
    T[:3, 3] = (0.05, -0.03, 1.2)
    Ts.append(T)
dm = ops.Mesh(mesh.vertices, mesh.faces, mesh.vertex_colors).set_ambient(5.0)
photos = [ops.rasterize(dm, torch.from_numpy(T[None].astype(np.float32)), 1.0, 600.0, 600.0, 320.0, 240.0, 640, 480)[0][0].cpu().numpy() for T in Ts]
tr = TrackingRefiner(dino_model="dinov2_vitb14_reg", seed=4)
res = {}
for window in (1, 16, 1, 16):
    tr.pose_confidences(mesh, photos[:2], K, Ts[:2], window=window)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res[window] = tr.pose_confidences(mesh, photos, K, Ts, window=window)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"window {window:2d}: {dt * 1e3:7.1f} ms for 32 (frame, pose) pairs = {dt / 32 * 1e3:.2f} ms per pair")
print("identical confidences:", np.array_equal(res[1], res[16]))
