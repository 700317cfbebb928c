"""per-kernel rasteriser timing probe (run under rocprofv3 --kernel-trace --stats)"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from freepose_amd import ops  # noqa: E402
from freepose_amd.src.pipeline.retrieval.renderer import grid_poses  # noqa: E402
sub, tiled = int(sys.argv[1]), int(sys.argv[2])
poses = torch.from_numpy(np.array(grid_poses(576)).astype(np.float32)).cuda()
v, f, c = bench.synthetic_mesh(sub)
m = ops.Mesh(v, f, c)
ops.set_option("raster_tiled", tiled)
for _ in range(5):
    ops.rasterize_extents(m, poses, 0.25, 600, 600, 210, 210, 420, 420)
torch.cuda.synchronize()
