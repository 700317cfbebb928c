"""K-tile ring depth of the 64x64 GEMM tier on small-batch ViT-L forwards (lab build): cap 8 (default) / 6 / 4 / 3 / 2.
python tools/ring_ab.py [res] [B,B,...]"""
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
for B in ([int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 3)):
    x = torch.rand((B, 3, res, res), device="cuda").to(torch.bfloat16)
    row = []
    for rounds in range(2):
        for cap in (8, 6, 4, 3, 2):
            ops.set_option("gemm_ring", cap)
            for _ in range(2):
                vit(x, layer=22, feature_type="patch")
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                t = ops.Timer(); t.start()
                vit(x, layer=22, feature_type="patch")
                t.stop(); ts.append(t.elapsed_ms())
            row.append(f"cap {cap}: {statistics.median(ts):.3f}")
    ops.set_option("gemm_ring", -1)
    print(f"B={B} @{res}: " + " | ".join(row), flush=True)
