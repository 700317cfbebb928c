"""ViT-L forward at a given batch under rocprofv3 (per-kernel-name totals).  B=6 RES=518 rocprofv3 --kernel-trace --stats -- python tools/vit_batch_prof.py"""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops
B, res, n = int(os.environ.get("B", "6")), int(os.environ.get("RES", "518")), int(os.environ.get("N", "5"))
if os.environ.get("SPLIT", "1") == "0":
    ops.set_option("gemm_row_split", 0)
vit = ops.ViT("dinov2_vitl14_reg", seed=0)
x = torch.rand((B, 3, res, res), device="cuda").to(torch.bfloat16)
for _ in range(n):
    vit(x, layer=22, feature_type="patch")
torch.cuda.synchronize()
t = ops.Timer(); t.start()
for _ in range(n):
    vit(x, layer=22, feature_type="patch")
t.stop()
print(f"B={B} res={res} split={os.environ.get('SPLIT', '1')}: {t.elapsed_ms() / n:.3f} ms per forward")
