"""Small-batch ViT-L forwards under dispatch policies of the GEMM small tier (lab build), same process, rotating order.
    python tools/small_policy_ab.py [res] [B,B,...]"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()
from freepose_amd import ops  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 420
vit = ops.ViT("dinov2_vitl14_reg", seed=0)
POL = (("HIP small tiers", 238), ("asm 128x128 wherever supported", 238 | 1048576), ("asm where every CU gets a tile", 238 | 1048576 | 2097152),
       ("asm for K >= 2048", 238 | 1048576 | 4194304), ("asm for K >= 2048 where every CU gets a tile", 238 | 1048576 | 4194304 | 2097152))
import os
if os.environ.get("POLICY") == "stream":     # big tier with / without non-temporal epilogue I/O (the asm big kernel always streams: bit 8192 keeps it out)
    POL = (("streaming epilogue I/O (product)", 238), ("plain stores on the big tier", 238 | 8388608), ("plain stores, no asm tier", 238 | 8388608 | 8192), ("no asm tier", 238 | 8192))
for B in ([int(b) for b in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2, 3, 4, 5, 6, 8, 12, 16, 21)):
    x = torch.rand((B, 3, res, res), device="cuda").to(torch.bfloat16)
    ts = {n: [] for n, _ in POL}
    for rnd in range(6):
        order = POL[rnd % len(POL):] + POL[:rnd % len(POL)]
        for name, var in order:
            ops.set_option("gemm_variant", var)
            vit(x, layer=22, feature_type="patch")
            torch.cuda.synchronize()
            t = ops.Timer(); t.start()
            for _ in range(3):
                vit(x, layer=22, feature_type="patch")
            t.stop(); ts[name].append(t.elapsed_ms() / 3)
    ops.set_option("gemm_variant", -1)
    base = statistics.median(ts[POL[0][0]])
    print(f"B={B} @{res}: " + " | ".join(f"{n} {statistics.median(v):.3f} ms ({statistics.median(v) / base:.3f})" for n, v in ts.items()), flush=True)
