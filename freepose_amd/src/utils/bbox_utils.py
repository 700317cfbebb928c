"""Drop-in for the reference's `src/utils/bbox_utils.py` (CropResizePad :9-56, bbox_iou :125-145, box helpers).

CropResizePad runs as one HIP kernel per call (fp_crop_resize_pad) instead of a per-box Python loop of
slice / F.interpolate / F.pad; the integer box arithmetic and torch's nearest-neighbour index rule are reproduced
bit for bit (tests/test_golden_cpu.py pins the rule, tests/test_gpu_retrieval.py the kernel).
"""
from __future__ import annotations

from typing import Tuple, Union

import numpy as np
import torch

from freepose_amd import ops


def unresizable_box(boxes, h: int, w: int, target: int, bbox_extend: float) -> int:
    """Index of the first box the reference cannot crop, or -1: after the extension and the clip to the image (bbox_utils.py:20-28) the
    crop is empty, or `F.interpolate(scale_factor = target / max side)` (:30-35) would have to produce a side of 0 px — torch raises
    "Input and output sizes should be greater than 0" there and the reference's script ends — or the crop does not come out at the
    target size (see below).  Same integer / float32 arithmetic as the
    device kernel (csrc/pose.hip crop_params_kernel) and the oracle (fpo_crop_resize_pad), on the host: no device round trip."""
    b = np.asarray(boxes).reshape(-1, 4).astype(np.int64)
    x0, y0, x1, y1 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    if float(bbox_extend) == 0.0:
        x0, y0, x1, y1 = np.maximum(x0, 0), np.maximum(y0, 0), np.minimum(x1, w), np.minimum(y1, h)
    else:
        e = np.float32(bbox_extend)
        ew, eh = e * (x1 - x0).astype(np.float32), e * (y1 - y0).astype(np.float32)
        fx0, fx1 = x0.astype(np.float32) - ew, x1.astype(np.float32) + ew
        fy0, fy1 = y0.astype(np.float32) - eh, y1.astype(np.float32) + eh
        x0 = np.where(fx0 > 0, np.trunc(fx0), 0).astype(np.int64)
        x1 = np.where(fx1 < np.float32(w), np.trunc(fx1), w).astype(np.int64)
        y0 = np.where(fy0 > 0, np.trunc(fy0), 0).astype(np.int64)
        y1 = np.where(fy1 < np.float32(h), np.trunc(fy1), h).astype(np.int64)
    cw, ch = x1 - x0, y1 - y0
    side = np.maximum(np.maximum(cw, ch), 1)
    scale = ((np.float32(1.0) / side.astype(np.float32)) * np.float32(target)).astype(np.float64)      # reciprocal(tensor) * scalar in float32, then .item()
    h1, w1 = np.floor(ch * scale), np.floor(cw * scale)
    # a crop that is exactly square is not padded to the target (:41-48) and the last resize (:52-54) can then come out one pixel short
    # (floor(h1 * (target / h1)) = target - 1): the reference's torch.stack of unequal crops, or the ViT's patch-size check, raises
    with np.errstate(divide="ignore", invalid="ignore"):
        s_h = np.where(w1 / h1 != 1.0, float(target), h1)
        out = np.floor(s_h * (float(target) / s_h))
    bad = np.flatnonzero((cw <= 0) | (ch <= 0) | (h1 <= 0) | (w1 <= 0) | (out != target))
    return int(bad[0]) if len(bad) else -1


class CropResizePad:
    def __init__(self, target_size: Union[Tuple, int], orig_size: Union[Tuple, int], bbox_extend: float = 0):
        if isinstance(target_size, int):
            target_size = (target_size, target_size)
        if target_size[0] != target_size[1]:
            # the reference cannot produce a non-square crop either: its own `assert image.shape[1] == image.shape[2]` after padding
            # (bbox_utils.py:47-49) fails for every non-square target, so there is no behaviour to reproduce
            raise NotImplementedError("non-square targets fail the reference's own square-after-padding assertion (bbox_utils.py:47-49)")
        self.target_size = tuple(target_size)
        self.target_h, self.target_w = self.target_size
        self.target_ratio = self.target_w / self.target_h
        self.target_max = max(self.target_size)
        self.bbox_extend = bbox_extend
        if isinstance(orig_size, int):
            orig_size = (orig_size, orig_size)
        self.h, self.w = orig_size

    def __call__(self, images: torch.Tensor, boxes: torch.Tensor, masks: torch.Tensor = None, mask_mode: int = 0,
                 out_bf16: bool = False) -> torch.Tensor:
        """images [n,C,H,W] float (one per box) or [1,C,H,W] shared, or u8 [n,H,W,C]; boxes [n,4] int xyxy.
        The clip limits are this object's (h, w) like the reference (:24-27); images are expected at that size."""
        boxes = torch.as_tensor(boxes)
        n = boxes.shape[0]
        if images.dtype != torch.uint8 and (images.shape[-2] != self.h or images.shape[-1] != self.w):
            raise ValueError(f"image size {tuple(images.shape[-2:])} differs from orig_size {(self.h, self.w)}")
        if images.shape[0] not in (1, n):
            raise ValueError("need one image per box (or a single shared image)")
        if not boxes.is_cuda:
            # the reference fails on such a box (torch raises inside F.interpolate, bbox_utils.py:35); boxes that live on the device (the
            # template loader's, cut from depth masks of >= 100 px or the fallback square) are not read back for this
            bad = unresizable_box(boxes.numpy(), self.h, self.w, self.target_max, float(self.bbox_extend))
            if bad >= 0:
                raise RuntimeError(f"CropResizePad: box {bad} {boxes[bad].tolist()} has an empty crop or resizes to a side of 0 px "
                                   f"(target {self.target_max}, extension {self.bbox_extend}) — the reference's F.interpolate raises here")
        return ops.crop_resize_pad(images, boxes, self.target_max, float(self.bbox_extend), masks, mask_mode, out_bf16)


def xyxy_to_xywh(bbox):
    bbox = np.asarray(bbox)
    if bbox.ndim == 1:
        return [bbox[0], bbox[1], bbox[2] - bbox[0] + 1, bbox[3] - bbox[1] + 1]
    if bbox.ndim == 2:
        return np.stack([bbox[:, 0], bbox[:, 1], bbox[:, 2] - bbox[:, 0], bbox[:, 3] - bbox[:, 1]], axis=1)
    raise ValueError("bbox must be a numpy array of shape (4,) or (N, 4)")


def xywh_to_xyxy(bbox):
    bbox = np.asarray(bbox)
    if bbox.ndim == 1:
        return [bbox[0], bbox[1], bbox[0] + bbox[2] - 1, bbox[1] + bbox[3] - 1]
    if bbox.ndim == 2:
        return np.stack([bbox[:, 0], bbox[:, 1], bbox[:, 0] + bbox[:, 2], bbox[:, 1] + bbox[:, 3]], axis=1)
    raise ValueError("bbox must be a numpy array of shape (4,) or (N, 4)")


def get_bbox_size(bbox):
    return [bbox[2] - bbox[0], bbox[3] - bbox[1]]


def bbox_iou(bb_a, bb_b):
    """IoU of two [x, y, w, h] boxes."""
    ax0, ay0, ax1, ay1 = bb_a[0], bb_a[1], bb_a[0] + bb_a[2], bb_a[1] + bb_a[3]
    bx0, by0, bx1, by1 = bb_b[0], bb_b[1], bb_b[0] + bb_b[2], bb_b[1] + bb_b[3]
    iw = min(ax1, bx1) - max(ax0, bx0)
    ih = min(ay1, by1) - max(ay0, by0)
    if iw > 0 and ih > 0:
        inter = iw * ih
        return inter / float(bb_a[2] * bb_a[3] + bb_b[2] * bb_b[3] - inter)
    return 0.0
