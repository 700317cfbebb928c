"""Secondary measurements (development): rasteriser across mesh sizes, ViT-L throughput at the bank-build and pose shapes."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from freepose_amd import ops  # noqa: E402
from freepose_amd.src.pipeline.retrieval.renderer import grid_poses  # noqa: E402


def timeit(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    t = ops.Timer()
    t.start()
    for _ in range(iters):
        fn()
    t.stop()
    return t.elapsed_ms() / iters


def main():
    poses = torch.from_numpy(np.array(grid_poses(576)).astype(np.float32)).cuda()
    for sub in (3, 4, 5, 6, 7):
        v, f, c = bench.synthetic_mesh(sub)
        m = ops.Mesh(v, f, c)
        res = []
        for tiled in (1, 0):
            ops.set_option("raster_tiled", tiled)
            res.append(timeit(lambda: ops.rasterize(m, poses, 0.25, 600, 600, 210, 210, 420, 420)))
        ops.set_option("raster_tiled", -1)
        ms = res[0]
        print(f"raster 576 views 420^2, {len(f):7d} triangles: tiled {ms:6.2f} ms ({576 * len(f) / ms / 1e6:6.1f} G tri/s, "
              f"{576 * 420 * 420 * 7 / ms / 1e6:5.0f} GB/s of mandatory writes) | global visibility buffer {res[1]:6.2f} ms", flush=True)
    vit = ops.ViT("dinov2_vitl14_reg", seed=0)
    for (B, H) in ((256, 420), (192, 518), (64, 518), (8, 518), (1, 518)):
        x = torch.rand((B, 3, H, H)).to(torch.bfloat16).cuda()
        ms = timeit(lambda: vit(x, layer=22, feature_type="patch"))
        fl = vit.flops(B, H, H, 22)
        print(f"ViT-L/14-reg layer 22, B={B:3d} @{H}^2: {ms:8.2f} ms  {B / ms * 1e3:7.1f} crops/s  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
