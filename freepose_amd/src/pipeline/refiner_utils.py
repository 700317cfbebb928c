"""Drop-in for the parts of the reference's `src/pipeline/refiner_utils.py` that `TrackingRefiner.pose_confidence` uses
(:26-50 tensor/normalise helpers, :92-132 crop_image, :135-170 update_K_with_crop).  The crop itself runs on the device
(`fp_roi_align`); the box arithmetic stays the reference's float32 torch expressions on a handful of points.

`cubic_resize` restates `cv2.resize(..., interpolation=cv2.INTER_CUBIC)` for float32 images (the 518->37 validity mask of
tracking_refiner.py:78): OpenCV is not installable here, so that piece is pinned by known-answer tests only.
"""
from __future__ import annotations

import numpy as np
import torch

from freepose_amd import ops

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


class MaybeToTensor:
    """torchvision ToTensor that passes tensors through (refiner_utils.py:26-31): HWC uint8 -> CHW float32 / 255"""

    def __call__(self, pic):
        if isinstance(pic, torch.Tensor):
            return pic
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


def pil2torch(pic) -> torch.Tensor:
    """MaybeToTensor + ImageNet normalise (refiner_utils.py:46-49)"""
    t = MaybeToTensor()(pic)
    mean = torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=t.dtype, device=t.device).view(3, 1, 1)
    std = torch.tensor(IMAGENET_DEFAULT_STD, dtype=t.dtype, device=t.device).view(3, 1, 1)
    return (t - mean) / std


def _project(points_h: torch.Tensor, P: torch.Tensor) -> torch.Tensor:
    """pixel coordinates of homogeneous points [n,4] under the 3x4 projections P [B,3,4]; depth clamped at 1 cm"""
    cam = torch.matmul(points_h.unsqueeze(0), P.permute(0, 2, 1))              # [B,n,3]
    return cam[:, :, :2] / torch.maximum(cam[:, :, [2]], torch.tensor(0.01))


def crop_boxes(Ts, points, K, render_width, render_height, lamb=1.4):
    """Crop box per pose, as the reference's crop_image derives it (refiner_utils.py:98-122): centred on the projection of
    the points' centroid, just large enough for every projected point, shaped like the render, scaled by `lamb`.
    float32 throughout, same operation order as the reference (pinned bit for bit by tests/golden/refiner.npz)."""
    assert Ts.shape[1:] == torch.Size([4, 4]) and points.shape[1:] == torch.Size([4]) and K.shape == torch.Size([3, 3])
    P = torch.matmul(torch.nn.functional.pad(K, (0, 1, 0, 0), value=0.).unsqueeze(0), Ts)     # K [I|0] T  -> [B,3,4]
    uv = _project(points, P)
    lo_hi = torch.cat([uv.min(dim=1).values, uv.max(dim=1).values], dim=1)                    # [B,4] = umin,vmin,umax,vmax
    centre = _project(torch.mean(points, dim=0, keepdim=True), P).squeeze(1)                  # [B,2]
    reach = torch.maximum((lo_hi[:, [0, 1]] - centre).abs_(), (lo_hi[:, [2, 3]] - centre).abs_())
    reach_x, reach_y = reach[:, 0], reach[:, 1]
    aspect = render_width / render_height
    box_w = torch.max(reach_x, reach_y * aspect) * 2 * lamb
    box_h = torch.max(reach_x / aspect, reach_y) * 2 * lamb
    half_w, half_h = box_w / 2, box_h / 2
    return torch.stack([centre[:, 0] - half_w, centre[:, 1] - half_h, centre[:, 0] + half_w, centre[:, 1] + half_h], dim=1)


def crop_image(image, Ts, points, K, render_width, render_height, lamb=1.4):
    """refiner_utils.py:92-132 — crop_boxes, then RoIAlign (sampling_ratio 2) of the image to (render_height, render_width)
    on the device"""
    assert len(image.shape) == 3 and image.shape[0] in [1, 3, 4] and image.dtype == torch.float32
    bboxes = crop_boxes(Ts, points, K, render_width, render_height, lamb)
    rois = torch.cat([torch.zeros((len(bboxes), 1)), bboxes], 1)
    crops = ops.roi_align(image.unsqueeze(0), rois, (render_height, render_width), sampling_ratio=2)
    return crops, bboxes


def update_K_with_crop(K, bboxes, render_width, render_height):
    """Intrinsics of the crop `bboxes` [B,4] (x1,y1,x2,y2) resampled to render_width x render_height, following the
    reference's convention (refiner_utils.py:135-170: pixel centres at integers, crop centre (w-1)/2, skew ignored)."""
    assert K.shape == torch.Size([3, 3]) and bboxes.shape[1:] == torch.Size([4])
    out = K.unsqueeze(0).repeat(len(bboxes), 1, 1)
    w, h = bboxes[:, 2] - bboxes[:, 0], bboxes[:, 3] - bboxes[:, 1]
    mid_x, mid_y = (bboxes[:, 0] + bboxes[:, 2]) / 2, (bboxes[:, 1] + bboxes[:, 3]) / 2
    # principal point inside the (unscaled) crop, then its offset from the crop centre
    off_x = (K[0, 2] + (w - 1) / 2 - mid_x) - (w - 1) / 2
    off_y = (K[1, 2] + (h - 1) / 2 - mid_y) - (h - 1) / 2
    sx, sy = render_width / w, render_height / h
    out[:, 0, 0] = sx * K[0, 0]
    out[:, 1, 1] = sy * K[1, 1]
    out[:, 0, 2] = (render_width - 1) / 2 + sx * off_x
    out[:, 1, 2] = (render_height - 1) / 2 + sy * off_y
    return out


def _cubic_coeffs(x: np.ndarray, A: float = -0.75) -> np.ndarray:
    """OpenCV interpolateCubic: 4 taps for the fractional offset x in [0,1)"""
    c = np.empty(x.shape + (4,), dtype=np.float32)
    x = x.astype(np.float32)
    c[..., 0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c[..., 1] = ((A + 2) * x - (A + 3)) * x * x + 1
    c[..., 2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    c[..., 3] = 1.0 - c[..., 0] - c[..., 1] - c[..., 2]
    return c


def cubic_resize(img: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(img float32 [H,W], dsize=(w,h), interpolation=cv2.INTER_CUBIC): half-pixel centres, a = -0.75, replicated
    border, separable (rows then columns) in float32"""
    src = np.asarray(img, dtype=np.float32)
    H, W = src.shape
    dw, dh = dsize

    def taps(n_src, n_dst):
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0).astype(np.float32)
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, _cubic_coeffs(frac)

    ix, cx = taps(W, dw)
    iy, cy = taps(H, dh)
    rows = (src[:, ix] * cx[None, :, :]).sum(-1, dtype=np.float32)          # [H, dw]
    out = (rows[iy, :] * cy[:, :, None]).sum(1, dtype=np.float32)           # [dh, dw]
    return out.astype(np.float32)
