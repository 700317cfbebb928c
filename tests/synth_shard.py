"""Deterministic synthetic template shard (test data builder: numpy + PIL only, no product or oracle imports).

Writes the reference's shard layout (scripts/render_templates.py:49-51,67-72): `shards/shard-000000.tar` with members
`<name>_<k>.rgb.png` (RGB u8 420x420) and `<name>_<k>.depth.png` (u16 millimetres), k in [0, 600), for two meshes, plus
`mesh_cache.csv` (column model_name).  Used twice: by oracle/gen_golden_r2.py, which feeds it to the REFERENCE's
WebTemplateDataset to produce tests/golden/template_dataset.npz, and by the GPU test that feeds the same bytes to the
mirror.  Views cover: ordinary blobs, off-centre / clipped blobs, masks below 100 px (the 210x210 fallback square,
template.py:75-77) and empty depth maps.
"""
from __future__ import annotations

import io
import tarfile
from pathlib import Path

import numpy as np
from PIL import Image

NAMES = ["0123456789abcdef0123456789abcdef", "Shark"]
N_VIEWS = 600
RES = 420


def view(mesh_idx: int, k: int):
    """(rgb u8 [420,420,3], depth u16 [420,420]) of view k"""
    rng = np.random.Generator(np.random.PCG64(1000 * (mesh_idx + 1) + k))
    yy, xx = np.mgrid[0:RES, 0:RES]
    depth = np.zeros((RES, RES), np.uint16)
    kind = k % 20
    if kind == 7:                       # empty view
        pass
    elif kind == 13:                    # tiny blob (< 100 px) -> fallback square
        cy, cx = int(rng.integers(20, 400)), int(rng.integers(20, 400))
        depth[cy:cy + 6, cx:cx + 9] = 1100
    else:
        cy, cx = rng.integers(60, 360, size=2)
        ry, rx = rng.integers(20, 190, size=2)
        if kind == 3:                   # clipped by the image border
            cy, cx = int(rng.integers(0, 40)), int(rng.integers(380, 420))
        m = ((yy - cy) / float(ry)) ** 2 + ((xx - cx) / float(rx)) ** 2 <= 1.0
        depth[m] = (900 + ((xx + 2 * yy) % 400))[m].astype(np.uint16)
    base = rng.integers(0, 256, size=(7, 7, 3), dtype=np.uint8)
    rgb = np.kron(base, np.ones((60, 60, 1), np.uint8))[:RES, :RES].copy()
    rgb[..., 0] = (rgb[..., 0].astype(np.int32) + xx) % 256
    rgb[..., 1] = (rgb[..., 1].astype(np.int32) + yy) % 256
    rgb[depth == 0] = 0
    return rgb, depth


def write_shard(root: Path, n_views: int = N_VIEWS):
    root = Path(root)
    (root / "shards").mkdir(parents=True, exist_ok=True)
    (root / "mesh_cache.csv").write_text("model_name\n" + "\n".join(NAMES) + "\n")
    with tarfile.open(root / "shards" / "shard-000000.tar", "w") as tar:
        for mi, name in enumerate(NAMES):
            for k in range(n_views):
                rgb, depth = view(mi, k)
                for suffix, img in (("rgb.png", Image.fromarray(rgb, "RGB")), ("depth.png", Image.fromarray(depth))):
                    buf = io.BytesIO()
                    img.save(buf, format="PNG", compress_level=1)
                    info = tarfile.TarInfo(f"{name.replace('_', '')}_{k}.{suffix}")
                    info.size = buf.tell()
                    buf.seek(0)
                    tar.addfile(info, buf)
    return [n.replace("_", "") for n in NAMES]
