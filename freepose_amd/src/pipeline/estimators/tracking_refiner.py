"""Drop-in for the GPU-bound part of the reference's `src/pipeline/estimators/tracking_refiner.py`: per-frame pose
confidence (`pose_confidence`, :70-93) and the inlier count used to pick trustworthy frames (`n_inliers_per_pose`,
`_get_threshold_for_confidence`, :60-68, :95-104).  Same two device kernels as the main path at other shapes — the ViT
(ViT-B/14-reg, 518^2, all 12 blocks + final norm = `x_norm_patchtokens`) and the rasteriser (518^2, cropped intrinsics,
ambient 5) — plus `fp_roi_align` for the photo crop.

Deviations, on purpose: features are computed in bf16 with fp32 accumulation (the reference runs this model in fp32 on
`dino_device`), the render comes from the HIP rasteriser instead of pyrender/EGL, and the CoTracker / PnP smoothing half of
the class (out of scope) raises NotImplementedError.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from freepose_amd import ops
from freepose_amd.mesh_io import device_mesh, mesh_signature
from freepose_amd.src.pipeline import refiner_utils


class TrackingRefiner:
    def __init__(self, dino_model="dinov2_vitb14_reg", dino_device="cuda", cotracker_device="cpu", state_dict=None, seed=0):
        self.dino_device = dino_device
        self.cotracker_device = cotracker_device
        self.dinov2 = ops.ViT(dino_model, state_dict=state_dict, seed=seed)
        self.patch_size = 14
        self.image_size = int(math.sqrt(1370 - 1) * self.patch_size)     # 518 (tracking_refiner.py:26)
        self.feats_size = self.image_size // self.patch_size              # 37
        self._mesh_cache = {}

    # ---- rendering / cropping --------------------------------------------------------------------------------------
    def _device_mesh(self, mesh) -> ops.Mesh:
        # keyed by identity AND content: the pipeline scales meshes in place (online_pose_estimator.py:60,64) and ids are recycled
        key, sig = id(mesh), mesh_signature(mesh)
        hit = self._mesh_cache.get(key)
        if hit is None or hit[1] != sig:
            hit = (device_mesh(mesh).set_ambient(5.0), sig)              # ambient_light=[5,5,5] (tracking_refiner.py:33)
            self._mesh_cache = {key: hit}
        return hit[0]

    def _render(self, mesh, width, height, K, transform):
        """colour u8 [H,W,3] and metric depth f32 [H,W] of `mesh` under `transform` (object -> OpenCV camera)"""
        K = np.asarray(K, dtype=np.float64)
        pose = torch.from_numpy(np.asarray(transform, dtype=np.float32).reshape(1, 4, 4))
        rgb, depth = ops.rasterize(self._device_mesh(mesh), pose, 1.0, K[0, 0], K[1, 1], K[0, 2], K[1, 2], width, height)
        return rgb[0], depth[0]

    @staticmethod
    def _sample_points(mesh) -> torch.Tensor:
        """100 homogeneous object points: the reference seeds the global NumPy generator with 42 and draws vertex indices
        with np.random.choice (:45-48); the same legacy stream without the global side effect"""
        vertices = np.asarray(mesh.vertices)
        pick = np.random.RandomState(42).choice(np.arange(len(vertices)), 100)
        return torch.from_numpy(np.pad(vertices[pick], ((0, 0), (0, 1)), constant_values=1.).copy()).float()

    def _crop_image(self, mesh, image, K, transform):
        vertices = self._sample_points(mesh)
        image = refiner_utils.MaybeToTensor()(image)
        K = torch.from_numpy(np.asarray(K)).view(3, 3).float()
        transform = torch.from_numpy(np.asarray(transform)).view(1, 4, 4).float()
        cropped_images, recomputed_bboxes = refiner_utils.crop_image(image, transform, vertices, K, 518, 518)
        new_Ks = refiner_utils.update_K_with_crop(K, recomputed_bboxes, 518, 518)
        return cropped_images[0], recomputed_bboxes[0], new_Ks[0]

    # ---- confidence ------------------------------------------------------------------------------------------------
    def _get_threshold_for_confidence(self, similarity_matrices, top_quantile=0.2):
        """lower edge of the histogram bin (50 bins over the positive similarities) at which the mass counted from the top
        first exceeds `top_quantile` of the total (tracking_refiner.py:60-68)"""
        counts, edges = np.histogram(similarity_matrices[similarity_matrices > 0], bins=50)
        budget = counts.sum() * top_quantile
        seen = 0
        edge = edges[0]
        for k in range(len(counts) - 1, -1, -1):
            seen += counts[k]
            edge = edges[k]
            if seen > budget:
                break
        return edge

    def _patch_features(self, chw_01: torch.Tensor) -> torch.Tensor:
        """x_norm_patchtokens of one image in [0,1] (ImageNet normalisation happens inside the ViT's im2col kernel)"""
        x = chw_01.unsqueeze(0).to(torch.bfloat16).cuda()
        f = self.dinov2(x, layer=len_blocks(self.dinov2), feature_type="patch")          # [1, 1369, D]
        return f[0].float()

    def pose_confidence(self, mesh, photo, K, transform):
        return self.pose_confidences(mesh, [photo], K, [transform])[0]

    def pose_confidences(self, mesh, frames, K, transforms, window=16):
        """pose_confidence (reference :70-96) for a list of (frame, pose) pairs: [n, 37, 37] masked patch cosines between the photo crop and
        the render of `mesh` under the pose.  The reference visits the pairs one by one with two B = 1 ViT-B forwards each
        (n_inliers_per_pose :98-103); they are independent, so the photo crops and renders of `window` pairs share ONE ViT call — a crop's
        features do not depend on its batch: the same confidences, pair for pair."""
        g = self.feats_size
        out = []
        pairs = list(zip(frames, transforms))
        for w0 in range(0, len(pairs), max(1, int(window))):
            chunk = pairs[w0:w0 + max(1, int(window))]
            crops, valids = [], []
            for photo, transform in chunk:
                cropped_photo, _new_bbox, new_K = self._crop_image(mesh, photo, K, transform)
                rendered, rendered_depth = self._render(mesh, 518, 518, new_K.numpy(), transform)
                valids.append(rendered_depth > 0)
                crops.append(cropped_photo.clamp(0, 1).to(torch.bfloat16))
                crops.append(rendered.permute(2, 0, 1).float().div(255).to(torch.bfloat16))
            feats = self.dinov2(torch.stack([c.cuda() for c in crops]), layer=len_blocks(self.dinov2), feature_type="patch").float()   # [2n, 1369, D]
            feats = feats / torch.linalg.norm(feats, dim=-1, keepdim=True)
            cos = (feats[0::2] * feats[1::2]).sum(-1).view(len(chunk), g, g).cpu()
            valid = torch.stack(valids).float().cpu().numpy()                                   # one device -> host copy per window
            for i in range(len(chunk)):
                render_valid_37x37_mask = refiner_utils.cubic_resize(valid[i], (37, 37)) > 0.5  # cv2.INTER_CUBIC (:78)
                out.append((cos[i] * torch.from_numpy(render_valid_37x37_mask).float()).numpy())
        return np.stack(out) if out else np.zeros((0, g, g), np.float32)

    def n_inliers_per_pose(self, mesh, frames, K, transforms):
        confidences = self.pose_confidences(mesh, frames, K, transforms)
        thr = self._get_threshold_for_confidence(confidences)
        return (confidences > thr).sum(-1).sum(-1), thr

    # ---- CoTracker / PnP smoothing: outside the hot path (out of scope) --------------------------------------
    def __getattr__(self, name):
        if name in ("refine", "refine_poses", "track", "_compute_3d_points", "smooth_poses"):
            raise NotImplementedError(f"TrackingRefiner.{name}: CoTracker/PnP smoothing is outside the MI355X hot path")
        raise AttributeError(name)


def len_blocks(vit: ops.ViT) -> int:
    return int(getattr(vit, "depth", 12))
