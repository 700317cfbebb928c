"""where the tiled rasteriser's time goes (LAB build: `raster_dbg` ablation bits of the tile kernel; wrong pictures, timing only)
     python tools/raster_ablate.py [subs=3,6]          1 = no rasterisation, 2 = no shading, 4 = no flush, 8 = no mask scan"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()
import bench  # noqa: E402
from freepose_amd import ops  # noqa: E402
from freepose_amd.src.pipeline.retrieval.renderer import grid_poses  # noqa: E402

subs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,6").split(",")]
poses = torch.from_numpy(np.array(grid_poses(576)).astype(np.float32)).cuda()


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
    m = ops.Mesh(v, f, c)
    ops.set_option("raster_tiled", 1)
    row = []
    for dbg, name in ((0, "all"), (2, "no shading"), (4, "no flush"), (1, "no raster"), (3, "no raster, no shading"), (7, "scan + LDS init only"),
                      (15, "LDS init only")):
        ops.set_option("raster_dbg", dbg)
        row.append(f"{name} {timed(lambda: ops.rasterize_extents(m, poses, 0.25, 600, 600, 210, 210, 420, 420)):.3f}")
    ops.set_option("raster_dbg", 0)
    print(f"{len(f):7d} triangles x 576 views [ms]: " + " | ".join(row), flush=True)

# ---- per-phase shader clocks of the tile kernel, summed over all workgroups (bit 30 of raster_dbg), and the per-lane / whole-wave threshold
import ctypes as C  # noqa: E402
lib = _lib.load()
lib.fp_lab_read_buffer.restype = C.c_int
lib.fp_lab_read_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
names = ["init", "mask scan", "chunk loop (all waves done)", "resolve", "flush", "-", "chunk loop of wave 0", "hit chunks"]
for sub in subs:
    v, f, c = bench.synthetic_mesh(sub)
    m = ops.Mesh(v, f, c)
    ops.set_option("raster_tiled", 1)
    for thr in (0, 4, 8, 16):
        ops.set_option("raster_dbg", (1 << 30) | (thr << 8))
        ops.rasterize_extents(m, poses, 0.25, 600, 600, 210, 210, 420, 420)
        buf = (C.c_ulonglong * 8)()
        _lib.check(lib.fp_lab_read_buffer(ops.context(), b"raster.dbg", buf, 64), "read")
        nwg = 49 * 576
        ops.set_option("raster_dbg", thr << 8)
        t = timed(lambda: ops.rasterize_extents(m, poses, 0.25, 600, 600, 210, 210, 420, 420))
        print(f"{len(f):7d} triangles, threshold {thr or 32:3d}: {t:.3f} ms | kilo-clocks per workgroup: " +
              ", ".join(f"{names[k]} {buf[k] / nwg / 1e3:.1f}" for k in (0, 1, 2, 3, 4, 6)) + f" | hit chunks per workgroup {buf[7] / nwg:.1f}", flush=True)
    ops.set_option("raster_dbg", 0)
