"""Drop-in for the reference's `src/pipeline/retrieval/renderer.py` (MeshRenderer :11-130).

The reference drives pyrender/OpenGL one pose at a time and reads every frame back to the host; here all poses of a
call are rasterised by the HIP kernels behind fp_rasterize (one launch set for the batch), and the renders stay on the
device: `render_from_poses` returns a `RenderBatch` that behaves like the reference's list of (rgb, depth, pose)
tuples when indexed, while `generate_proposals` consumes it without any device->host copy (fused mask bbox +
nearest crop/resize/pad kernels).

Camera / shading conventions restated from renderer.py: K = (600, 600, res/2, res/2) (:37), OpenCV camera frame
(:39-41), black background + ambient (2,2,2) only (:53-55), no face culling (:66).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from freepose_amd import ops
from freepose_amd.mesh_io import device_mesh, mesh_signature
from freepose_amd.src.utils.bbox_utils import CropResizePad


def super_fibonacci_rotations(n: int) -> np.ndarray:
    """[n,3,3] float64 — the SO(3) spiral of renderer.py:12-31 / pose_estimator.py:121-140."""
    s = np.arange(n, dtype=np.float64) + 0.5
    r, R = np.sqrt(s / n), np.sqrt(1.0 - s / n)
    alpha = 2.0 * np.pi * s / np.sqrt(2.0)
    beta = 2.0 * np.pi * s / 1.533751168755204288118041
    x, y, z, w = r * np.sin(alpha), r * np.cos(alpha), R * np.sin(beta), R * np.cos(beta)
    nn = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / nn, y / nn, z / nn, w / nn
    M = np.empty((n, 3, 3))
    M[:, 0, 0] = x * x - y * y - z * z + w * w
    M[:, 0, 1] = 2 * (x * y - z * w)
    M[:, 0, 2] = 2 * (x * z + y * w)
    M[:, 1, 0] = 2 * (x * y + z * w)
    M[:, 1, 1] = -x * x + y * y - z * z + w * w
    M[:, 1, 2] = 2 * (y * z - x * w)
    M[:, 2, 0] = 2 * (x * z - y * w)
    M[:, 2, 1] = 2 * (y * z + x * w)
    M[:, 2, 2] = -x * x - y * y + z * z + w * w
    return M


def grid_poses(n: int) -> List[np.ndarray]:
    Rs = super_fibonacci_rotations(n)
    out = []
    for i in range(n):
        P = np.eye(4)
        P[:3, :3] = Rs[i]
        P[:3, 3] = (0.0, 0.0, 1.1)
        out.append(P)
    return out


class RenderBatch(Sequence):
    """Device-resident renders; `batch[i]` -> (rgb uint8 [H,W,3], depth float32 [H,W], third) as numpy, like the
    reference's result tuples (third = pose for render_from_poses, rotation for render)."""

    def __init__(self, rgb: torch.Tensor, depth: torch.Tensor, thirds, intrinsics, extents=None, boxes=None):
        self.rgb, self.depth, self.thirds, self.intrinsics = rgb, depth, list(thirds), intrinsics
        # what fp_depth_extents would give on `depth` (f64 [n,8]) and its box columns (i32 [n,4]): written by the rasteriser's tile
        # epilogue (fp_rasterize_extents), so generate_proposals needs no pass over the depth images — `depth` itself may be None
        self.extents, self.boxes = extents, boxes

    def __len__(self):
        return self.rgb.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = int(i)
        if self.depth is None:
            raise RuntimeError("this batch was rendered without its depth images (render_from_poses(..., depth=False))")
        return self.rgb[i].cpu().numpy(), self.depth[i].cpu().numpy(), self.thirds[i]


class MeshRenderer:
    def __init__(self, n_poses, resolution=420):
        self.resolution = resolution
        self.mesh_poses = grid_poses(n_poses)
        self.rotations = [P[:3, :3] for P in self.mesh_poses]
        self.fx = self.fy = 600.0
        self.cx = self.cy = resolution / 2
        self.opencv2opengl = np.diag([1.0, -1.0, -1.0, 1.0])  # kept for API parity; the rasteriser works in the OpenCV frame
        self._mesh_cache = {}
        self.mesh_cache_size = 8

    def _device_mesh(self, mesh) -> ops.Mesh:
        if isinstance(mesh, ops.Mesh):
            return mesh
        key, sig = id(mesh), mesh_signature(mesh)
        hit = self._mesh_cache.get(key)
        if hit is not None and hit[1] == sig:
            self._mesh_cache[key] = self._mesh_cache.pop(key)         # most recently used last
            return hit[0]
        dm = device_mesh(mesh)
        self._mesh_cache.pop(key, None)
        self._mesh_cache[key] = (dm, sig)
        while len(self._mesh_cache) > self.mesh_cache_size:           # the objects of a video frame alternate: keep a few resident
            self._mesh_cache.pop(next(iter(self._mesh_cache)))
        return dm

    def _render(self, mesh, poses, thirds, scale=1.0, cull_faces=False, depth=True) -> RenderBatch:
        poses = np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4)
        dm = self._device_mesh(mesh)
        before = getattr(dm, "cull", 0)
        dm.set_cull(1 if cull_faces else 0)          # reference :63-66 / :90-93: SKIP_CULL_FACES unless cull_faces
        try:
            rgb, dimg, ext, boxes = ops.rasterize_extents(dm, torch.from_numpy(poses), scale, self.fx, self.fy, self.cx,
                                                          self.cy, self.resolution, self.resolution, want_depth=depth)
        finally:
            dm.set_cull(before)                       # a caller's own ops.Mesh keeps the mode it had
        return RenderBatch(rgb, dimg, thirds, (self.fx, self.fy, self.cx, self.cy), ext, boxes)

    def render(self, mesh, cull_faces=False, scale=1.0):
        return self._render(mesh, self.mesh_poses, self.rotations, scale, cull_faces)

    def render_from_poses(self, mesh, poses, cull_faces=False, scale=1.0, depth=True):
        """`depth=False`: the depth images are not written (their box / extents still are: RenderBatch.extents) — for callers that go
        straight to generate_proposals(..., need_masks=False)"""
        poses = list(poses)
        return self._render(mesh, poses, poses, scale, cull_faces, depth)

    @staticmethod
    def mask_to_bbox(mask):
        ys, xs = np.nonzero(mask)
        return np.array([xs.min(), ys.min(), xs.max(), ys.max()])

    @staticmethod
    def generate_proposals(res, resolution=420, bbox_extend=0, out_bf16=False, return_extents=False, need_masks=True):
        """(crops [n,3,res,res], poses, masks) from renders; masks = depth > 0 with the <100 px fallback square.  `need_masks=False`
        (the video step without mask_scores) skips the mask tensors — and with them a host synchronisation per call."""
        if not isinstance(res, RenderBatch):  # reference-style list of numpy tuples
            rgb = torch.from_numpy(np.stack([r[0] for r in res])).cuda()
            depth = torch.from_numpy(np.stack([r[1] for r in res]).astype(np.float32)).cuda()
            res = RenderBatch(rgb, depth, [r[2] for r in res], (600.0, 600.0, 210.0, 210.0))
        fx, fy, cx, cy = res.intrinsics
        if res.extents is not None:                   # from the rasteriser's tile epilogue: the same bits
            ext, boxes = res.extents, res.boxes
        else:
            ext = ops.depth_extents(res.depth, fx, fy, cx, cy)
            boxes = ext[:, :4].to(torch.int32)
        crops = ops.crop_resize_pad(res.rgb, boxes, resolution, float(bbox_extend), out_bf16=out_bf16)
        masks = None
        if need_masks:
            if res.depth is None:
                raise RuntimeError("masks need the depth images: render with depth=True")
            masks = res.depth > 0
            small = ext[:, 6] < 100
            if bool(small.any()):
                masks = masks.clone()
                masks[small, 105:315, 105:315] = True
        if return_extents:
            return crops, res.thirds, masks, ext
        return crops, res.thirds, masks
