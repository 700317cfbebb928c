"""BOP-format image index feeding the static-image driver (reference: src/dataloader/base_bop.py:11-106 and
bop.py:19-65; SURVEY §2 row 15 — plain file I/O, outside the accelerated path, kept so the CLI is a drop-in).
Per item: image (uint8 HxWx3), scene_id, frame_id, intrinsic (float32 3x3); depth maps only if present."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pandas as pd
from PIL import Image


class BOPDataset:
    def __init__(self, root_dir: str, split: str, use_visible_masks: bool = True, **kwargs):
        if "tless" in root_dir or "hb" in root_dir:
            split = f"{split}_primesense"
        self.path = Path(root_dir).resolve()
        self.split = split
        # own cache file: the reference writes `<split>_metadata.json` with a different schema (base_bop.py:60-106) and would
        # fail on ours with a KeyError (and vice versa)
        meta = self.path / f"{split}_metadata_fp.json"
        self.meta_data = pd.read_json(meta) if meta.exists() else self._index(meta)

    def _index(self, meta_path: Path) -> pd.DataFrame:
        rows = []
        for scene in sorted((self.path / self.split).iterdir()):
            cam = json.loads((scene / "scene_camera.json").read_text())
            for rgb in sorted(list(scene.glob("rgb/*.png")) + list(scene.glob("rgb/*.jpg")) + list(scene.glob("rgb/*.tif"))):
                fid = int(rgb.stem)
                depth = next(iter(sorted(scene.glob(f"depth/{rgb.stem}.*"))), None)
                rows.append({"scene_id": scene.name, "frame_id": fid, "rgb_path": str(rgb),
                             "depth_path": None if depth is None else str(depth), "intrinsic": cam[str(fid)]["cam_K"]})
        df = pd.DataFrame(rows)
        try:
            meta_path.write_text(df.to_json())
        except OSError:
            pass
        return df

    def __len__(self):
        return len(self.meta_data)

    def frame_key(self, idx):
        """(scene_id, frame_id) of entry idx without reading its image (the CLI looks one image ahead in the proposals file)"""
        row = self.meta_data.iloc[idx]
        return int(row["scene_id"]), int(row["frame_id"])

    def __getitem__(self, idx):
        row = self.meta_data.iloc[idx]
        image = np.asarray(Image.open(row["rgb_path"]).convert("RGB")).copy()
        out = {"image": image, "scene_id": row["scene_id"], "frame_id": row["frame_id"],
               "intrinsic": np.asarray(row["intrinsic"]).reshape(3, 3).astype(np.float32)}
        dp = row.get("depth_path")
        if isinstance(dp, str) and Path(dp).exists():
            out["depth"] = (np.asarray(Image.open(dp)).copy() * 0.1) / 1000
        return out
