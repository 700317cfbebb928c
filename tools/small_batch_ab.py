"""Small-batch ViT-L forwards under the GEMM tile tiers (lab build): default dispatch (64x64 tier on a 4-deep K-tile ring), the 64x64
tier on two K-tile buffers (bit 131072: the form shipped until round 4), 64x64 tiles for everything below the big tier (bit 32768),
128x128 kept for small grids (bit 2048).  python tools/small_batch_ab.py [res] [B,B,...]"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()
from freepose_amd import ops  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 518
vit = ops.ViT("dinov2_vitl14_reg", seed=0)
for B in ([int(b) for b in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2, 3, 4, 5, 6, 8, 12, 21)):
    x = torch.rand((B, 3, res, res), device="cuda").to(torch.bfloat16)
    row = []
    for name, var in (("default", -1), ("two K-tile buffers", 238 | 131072), ("64x64 forced below the big tier", 238 | 32768), ("no 64x64", 238 | 2048)):
        ops.set_option("gemm_variant", var)
        for _ in range(2):
            vit(x, layer=22, feature_type="patch")
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = ops.Timer(); t.start()
            vit(x, layer=22, feature_type="patch")
            t.stop(); ts.append(t.elapsed_ms())
        ms = statistics.median(ts)
        row.append(f"{name} {ms:.3f} ms ({vit.flops(B, res, res, 22) / ms / 1e9:.0f} TF)")
    ops.set_option("gemm_variant", -1)
    print(f"B={B} @{res}: " + " | ".join(row), flush=True)
