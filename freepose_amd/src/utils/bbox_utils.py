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
