"""Drop-in for the reference's `scripts/render_templates.py` (:27-75): render the 600 template views of every mesh of a
SLURM slice (10 meshes per task) and write them as `shard-%06d.tar` in the webdataset layout the template loader reads
(`<key>_<i>.rgb.png` uint8 RGB, `<key>_<i>.depth.png` uint16 millimetres; key = mesh id without underscores).

Differences, on purpose: the views come from the HIP rasteriser (`MeshRenderer`, no pyrender / EGL context), all 600 views of
a mesh are rendered in one device batch, and the tar is written with the standard library (webdataset is not needed to
produce its format).  `--n_views` / `--resolution` exist for tests; the defaults are the reference's 600 x 420^2.
"""
from __future__ import annotations

import argparse
import io
import os
import tarfile
import time
from pathlib import Path

import numpy as np
from PIL import Image

from freepose_amd.mesh_io import load_obj
from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer

MESHES_PER_SHARD = 10          # render_templates.py:42-44


def _png(arr: np.ndarray) -> bytes:
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="PNG")
    return b.getvalue()


def _add(tar: tarfile.TarFile, name: str, data: bytes, mtime: float):
    ti = tarfile.TarInfo(name)
    ti.size = len(data)
    ti.mtime = mtime
    tar.addfile(ti, io.BytesIO(data))


def write_shard(tar_path: Path, meshes, renderer: MeshRenderer, scale: float = 0.25, log=print) -> int:
    """meshes: iterable of (mesh_id, mesh object accepted by MeshRenderer).  Returns the number of views written."""
    n = 0
    now = time.time()
    with tarfile.open(tar_path.as_posix(), "w") as tar:
        for idx, (mesh_id, mesh) in enumerate(meshes):
            log(f"Rendering mesh {mesh_id} ({idx + 1})")
            mesh.apply_scale(scale)                                    # render_templates.py:62
            batch = renderer.render(mesh, cull_faces=False)
            rgb = batch.rgb.cpu().numpy()                              # [T,H,W,3] u8
            depth_mm = (batch.depth.cpu().numpy() * 1000).astype(np.uint16)   # :72, truncating like the reference
            key = mesh_id.replace("_", "")
            for i in range(rgb.shape[0]):
                _add(tar, f"{key}_{i}.rgb.png", _png(rgb[i]), now)
                _add(tar, f"{key}_{i}.depth.png", _png(depth_mm[i]), now)
                n += 1
    return n


def run(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--filelist", type=str, default="./data/mesh_cache.txt")
    ap.add_argument("--shards_folder", type=str, default="objaverse_shards")
    ap.add_argument("--offset", type=int, default=0)
    ap.add_argument("--mesh_root", type=str, default="data/mesh_cache")
    ap.add_argument("--datasets_root", type=str, default="./data/datasets")
    ap.add_argument("--n_views", type=int, default=600)
    ap.add_argument("--resolution", type=int, default=420)
    args = ap.parse_args(argv)

    shards_path = Path(args.datasets_root).resolve() / args.shards_folder
    shards_path.mkdir(parents=True, exist_ok=True)
    with open(args.filelist, "r") as f:
        mesh_ids = f.read().splitlines()
    job_id = int(os.getenv("SLURM_ARRAY_TASK_ID", 0)) + args.offset          # :39-41
    ids = mesh_ids[job_id * MESHES_PER_SHARD:(job_id + 1) * MESHES_PER_SHARD]
    renderer = MeshRenderer(args.n_views, resolution=args.resolution)
    root = Path(args.mesh_root).resolve()
    meshes = ((m, load_obj(root / m / f"{m}.obj")) for m in ids)
    tar_path = shards_path / f"shard-{job_id:06d}.tar"
    n = write_shard(tar_path, meshes, renderer)
    print(f"wrote {n} views of {len(ids)} meshes to {tar_path}")
    return tar_path


if __name__ == "__main__":
    run()
