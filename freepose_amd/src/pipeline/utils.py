"""Drop-in for the on-path parts of the reference's `src/pipeline/utils.py`:
`Proposals` (:18-69), `depthmap_to_pointcloud` (:122-145), `get_z_from_pointcloud` (:148-170), `mask_to_bbox`
(:172-181), plus the proposals-JSON mask codec (`mask_to_rle_pytorch` / `rle_to_mask`, vendored
sam2/utils/amg.py:109-151) as plain numpy.

Proposals crops every detection with ONE HIP launch per tensor (fp_crop_resize_pad: fused u8->float /255, mask
multiply, box extension, nearest crop/resize/pad) instead of repeating the image per proposal on the host.
"""
from __future__ import annotations

from typing import Any, Dict, List

import numpy as np
import torch

from freepose_amd import ops
from freepose_amd.src.utils.bbox_utils import CropResizePad, unresizable_box


# ---- proposals-JSON mask codec: uncompressed COCO RLE, column-major, first run = background -----------------
def mask_to_rle_pytorch(tensor) -> List[Dict[str, Any]]:
    m = np.asarray(tensor.cpu() if hasattr(tensor, "cpu") else tensor).astype(bool)
    out = []
    for mask in m:
        h, w = mask.shape
        flat = mask.T.reshape(-1)                      # Fortran order
        change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
        edges = np.concatenate([[0], change, [h * w]])
        runs = np.diff(edges).tolist()
        counts = ([0] if flat[0] else []) + runs
        out.append({"size": [h, w], "counts": counts})
    return out


def rle_to_mask(rle: Dict[str, Any]) -> np.ndarray:
    h, w = rle["size"]
    counts = np.asarray(rle["counts"], dtype=np.int64)
    vals = (np.arange(len(counts)) & 1).astype(bool)   # runs alternate, starting with background
    flat = np.repeat(vals, counts)
    if flat.size != h * w:
        raise ValueError("RLE does not cover the mask")
    return flat.reshape(w, h).T


class Proposals:
    """image: uint8 [H,W,3]; detections_output: {'masks': bool [n,H,W], 'boxes': int [n,4] xyxy}."""

    def __init__(self, image, detections_output, target_size=350, scene_id=None, frame_id=None, bbox_extend=0.2,
                 mask_rgb=True):
        arr = np.ascontiguousarray(image)
        if not arr.flags.writeable:                      # (np.asarray of a PIL image: torch wants a writable buffer)
            arr = arr.copy()
        self._image_u8 = torch.as_tensor(arr, dtype=torch.uint8)
        self.masks = torch.as_tensor(detections_output["masks"]).bool()
        self.boxes = torch.as_tensor(detections_output["boxes"]).int()
        self.rgb_proposal_processor = CropResizePad(target_size=target_size, orig_size=(image.shape[0], image.shape[1]),
                                                    bbox_extend=bbox_extend)
        self.proposals, self.proposals_masks = self.extract_proposals(mask_rgb=mask_rgb)
        self.features = None
        self.scores = []
        self.meshes = []
        self.scene_id = scene_id
        self.frame_id = frame_id

    @property
    def image(self) -> torch.Tensor:
        """float CHW image in [0,1] (reference attribute; not used on the fast path)."""
        return (self._image_u8.float() / 255).permute(2, 0, 1)

    def extract_proposals(self, mask_rgb=True):
        n = len(self.masks)
        T = self.rgb_proposal_processor.target_max
        if n == 0:
            return torch.empty(0, 3, T, T), torch.empty(0, T, T, dtype=torch.bool)
        img = self._image_u8[None].cuda()
        masks = self.masks.cuda().to(torch.uint8)
        ext = float(self.rgb_proposal_processor.bbox_extend)
        bad = unresizable_box(self.boxes.cpu().numpy(), self._image_u8.shape[0], self._image_u8.shape[1], T, ext)
        if bad >= 0:           # the reference's CropResizePad raises on such a detection (torch, bbox_utils.py:35)
            raise RuntimeError(f"Proposals: detection {bad} {self.boxes[bad].tolist()} has an empty crop or resizes to a side of 0 px")
        rgbs = ops.crop_resize_pad(img, self.boxes, T, ext, masks, 1 if mask_rgb else 0, u8_float_div=True)
        m = ops.crop_resize_pad(img, self.boxes, T, ext, masks, 2, u8_float_div=True)
        return rgbs, m[:, 0] > 0.5

    def to_bop_dict(self):
        boxes = self.boxes.cpu().numpy()
        rles = mask_to_rle_pytorch(self.masks)
        out = []
        for i in range(len(boxes)):
            x0, y0, x1, y1 = (int(v) for v in boxes[i])
            out.append({"bbox": [x0, y0, x1 - x0, y1 - y0], "segmentation": rles[i], "mesh": self.meshes[i],
                        "score": self.scores[i], "scene_id": int(self.scene_id), "image_id": int(self.frame_id),
                        "time": 0.01})
        return out


# ---- depth -> metric pose helpers (float64 numpy, host logic) -------------------------------------------------
def depthmap_to_pointcloud(depth_map, K):
    """Back-project every pixel with K^-1, drop rows that are exactly zero (background)."""
    Kinv = np.linalg.inv(np.asarray(K))
    h, w = depth_map.shape[:2]
    u, v = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    pix = np.stack((u, v, np.ones_like(u)), axis=2).reshape(-1, 3)
    pts = (np.dot(Kinv, pix.T) * depth_map.flatten()).T
    return pts[~np.all(pts == 0, axis=1)]


def z_from_extents(bbox, dx3d, dy3d, K, TCO_init):
    """get_z_from_pointcloud on precomputed cloud extents (what the fused HIP extents kernel returns)."""
    TCO = np.array(TCO_init, dtype=np.float64, copy=True)
    K = np.asarray(K)
    bbox = np.asarray(bbox.cpu() if hasattr(bbox, "cpu") else bbox)
    fxfy = K[[0, 1], [0, 1]]
    cxcy = K[[0, 1], [2, 2]]
    centre = (bbox[0:2] + bbox[2:4]) / 2
    z = (fxfy[1] * dy3d / ((bbox[3] - bbox[1]) + 1) + fxfy[0] * dx3d / ((bbox[2] - bbox[0]) + 1)) / 2
    TCO[:2, 3] = ((centre - cxcy) * z) / fxfy
    TCO[2, 3] = z
    return TCO


def get_z_from_pointcloud(bbox, pointcloud, K, TCO_init):
    dx = pointcloud[:, 0].max() - pointcloud[:, 0].min()
    dy = pointcloud[:, 1].max() - pointcloud[:, 1].min()
    return z_from_extents(bbox, dx, dy, K, TCO_init)


def mask_to_bbox(mask):
    ys, xs = np.nonzero(mask)
    return np.array([xs.min(), ys.min(), xs.max(), ys.max()])
