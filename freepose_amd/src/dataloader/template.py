"""Drop-in for the reference's `src/dataloader/template.py` (WebTemplateDataset :26-99): random-access loader of the
pre-rendered template shards (10 meshes per `shard-%06d.tar`, 600 x (rgb.png, depth.png u16 mm) per mesh, member index
cached beside the tar).

Host side: tar seek + PNG decode (PIL).  Device side: depth>0 masks, bounding boxes (fp_depth_extents) and the 600
crops (fp_crop_resize_pad) — the reference spends ~33 s per mesh in a Python crop loop.
"""
from __future__ import annotations

import io
import tarfile
from pathlib import Path

import numpy as np
import pandas as pd
import torch
from PIL import Image

from freepose_amd import ops
from freepose_amd.src.utils.bbox_utils import CropResizePad

N_VIEWS = 600


class WebTemplateDataset:
    def __init__(self, wds_dir: str, filelist_path: str, resolution: int = 420, bbox_extend: float = 0, crop: bool = True,
                 n_views: int = N_VIEWS):
        self.wds_dir = Path(wds_dir).resolve()
        self.frame_index = pd.read_csv(Path(filelist_path).resolve(), dtype=str)["model_name"].str.replace("_", "")
        self.rgb_proposal_processor = CropResizePad(resolution, (420, 420), bbox_extend=bbox_extend)
        self.resolution = resolution
        self.crop = crop
        self.n_views = n_views

    def __len__(self):
        return len(self.frame_index)

    def get_template_by_name(self, model_name):
        idx = self.frame_index[self.frame_index == model_name].index[0]
        return self.__getitem__(idx)

    def _member_index(self, tar, tar_path: Path):
        side = tar_path.with_suffix(".npy")
        if side.exists():
            return np.load(side, allow_pickle=True).item()
        index = {m.name: m for m in tar.getmembers()}
        try:
            np.save(side, index, allow_pickle=True)
        except OSError:
            pass
        return index

    def __getitem__(self, idx: int):
        shard = idx // 10
        tar_path = self.wds_dir / f"shard-{shard:06d}.tar"
        name = self.frame_index[idx].replace("_", "")
        with tarfile.open(tar_path.as_posix()) as tar:
            members = self._member_index(tar, tar_path)
            rgbs, depths = [], []
            for k in range(self.n_views):
                rgb = Image.open(io.BytesIO(tar.extractfile(members[f"{name}_{k}.rgb.png"]).read())).convert("RGB")
                dep = Image.open(io.BytesIO(tar.extractfile(members[f"{name}_{k}.depth.png"]).read()))
                rgbs.append(np.asarray(rgb, dtype=np.uint8))
                depths.append(np.asarray(dep))
        if not rgbs:
            return {"templates": None, "masks": None, "depths": None, "bboxes": None, "model_name": name, "tar_file": tar_path.name}
        rgb_u8 = torch.from_numpy(np.stack(rgbs)).cuda()                                  # [T,H,W,3] u8
        depth = torch.from_numpy((np.stack(depths) / 1000).astype(np.float32)).cuda()     # metres, float32 (:72)
        ext = ops.depth_extents(depth, 600.0, 600.0, 210.0, 210.0)
        masks = depth > 0
        small = ext[:, 6] < 100
        if bool(small.any()):
            masks[small, 105:315, 105:315] = True
        boxes = ext[:, :4].to(torch.int32)
        if self.crop:
            templates = ops.crop_resize_pad(rgb_u8, boxes, self.resolution, float(self.rgb_proposal_processor.bbox_extend))
        else:
            templates = ops.crop_resize_pad(rgb_u8, torch.tensor([[0, 0, rgb_u8.shape[2], rgb_u8.shape[1]]] * len(rgbs),
                                                                dtype=torch.int32), rgb_u8.shape[1], 0.0)
        intrinsic = torch.tensor([[600, 0, 210], [0, 600, 210], [0, 0, 1]]).reshape(3, 3)
        return {"templates": templates, "masks": masks, "depths": depth, "model_name": name, "tar_file": tar_path.name,
                "intrinsic": intrinsic, "bboxes": boxes}
