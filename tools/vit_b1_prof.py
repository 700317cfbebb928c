"""B = 1 ViT-L forward kernel breakdown (run under rocprofv3 --kernel-trace --stats)"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
vit = ops.ViT("dinov2_vitl14_reg", seed=0)
x = torch.rand((B, 3, 518, 518)).to(torch.bfloat16).cuda()
for _ in range(3):
    vit(x, layer=22, feature_type="patch")
torch.cuda.synchronize()
t = ops.Timer(); t.start()
for _ in range(20):
    vit(x, layer=22, feature_type="patch")
t.stop()
print(f"B={B}: {t.elapsed_ms() / 20:.3f} ms per forward", flush=True)
