"""rasteriser stage timings (HIP events): 576 views @420^2 of the bench's displaced icosphere at several triangle counts, both
visibility strategies, plain and fused (extents from the tile epilogue, no depth image), faces in mesh order and shuffled; then the
crop stage.      python tools/raster_perf.py [subs=3,5,6,7] [views=576]"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from freepose_amd import ops  # noqa: E402
from freepose_amd.src.pipeline.retrieval.renderer import grid_poses  # noqa: E402

subs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,5,6,7").split(",")]
n_views = int(sys.argv[2]) if len(sys.argv) > 2 else 576
poses = torch.from_numpy(np.array(grid_poses(n_views)).astype(np.float32)).cuda()


def timed(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


for sub in subs:
    v, f, c = bench.synthetic_mesh(sub)
    for order in ("mesh", "shuffled"):
        ff = f if order == "mesh" else np.ascontiguousarray(f[np.random.default_rng(1).permutation(len(f))])
        m = ops.Mesh(v, ff, c)
        row = []
        for tiled in (1, 0):
            ops.set_option("raster_tiled", tiled)
            t_plain = timed(lambda: ops.rasterize(m, poses, 0.25, 600, 600, 210, 210, 420, 420))
            t_ext = timed(lambda: ops.depth_extents(ops.rasterize(m, poses, 0.25, 600, 600, 210, 210, 420, 420)[1], 600, 600, 210, 210))
            t_fused = timed(lambda: ops.rasterize_extents(m, poses, 0.25, 600, 600, 210, 210, 420, 420))
            row.append(f"{'tiled' if tiled else 'global'}: rgb+depth {t_plain:.3f}  +extents kernel {t_ext:.3f}  fused, no depth {t_fused:.3f}")
        ops.set_option("raster_tiled", -1)
        print(f"{len(ff):7d} triangles ({order:8s}) x {n_views} views [ms]  " + "   |   ".join(row), flush=True)
v, f, c = bench.synthetic_mesh(6)
m = ops.Mesh(v, f, c)
rgb, _, ext, boxes = ops.rasterize_extents(m, poses, 0.25, 600, 600, 210, 210, 420, 420)
for res in (420, 518):
    t = timed(lambda: ops.crop_resize_pad(rgb, boxes, res, 0.0, out_bf16=True))
    by = n_views * 3 * res * res * 2
    print(f"crop_resize_pad {n_views} x {res}^2 bf16: {t:.3f} ms = {by / t / 1e6:.0f} GB/s of output", flush=True)
# FFA of one crop and of a bank-build batch (ViT-L @518^2: 1369 patches x 1024)
feats = torch.randn(256, 1369, 1024, device="cuda").to(torch.bfloat16)
masks = (torch.rand(256, 518, 518, device="cuda") > 0.6)
for B in (1, 5, 256):
    t = timed(lambda: ops.ffa(feats[:B], masks[:B], cell=14, normalize=True), it=20)
    print(f"ffa (cell mask + masked mean + normalise) B={B}: {t * 1e3:.1f} us = {B * 1369 * 1024 * 2 / t / 1e6:.0f} GB/s", flush=True)
