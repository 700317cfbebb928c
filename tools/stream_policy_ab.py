"""Streaming policy of the big GEMM tier (lab build): ViT-L forwards with the output-size threshold above which the 256x256 kernels store
non-temporally set to 0 (always stream: rounds 1-4), 128, 256, 512 MiB and never.  python tools/stream_policy_ab.py [res] [B,B,...]"""
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
POL = (("always stream", 0), ("> 128 MiB", 128), ("> 256 MiB", 256), ("> 512 MiB", 512), ("never", 1 << 20))
for B in ([int(b) for b in sys.argv[2].split(',')] if len(sys.argv) > 2 else (5, 12, 21, 32, 48, 64, 96, 144, 288)):
    x = torch.rand((B, 3, res, res), device="cuda").to(torch.bfloat16)
    ts = {n: [] for n, _ in POL}
    for rnd in range(5):
        order = POL[rnd % len(POL):] + POL[:rnd % len(POL)]
        for name, mb in order:
            ops.set_option("gemm_stream_mb", mb)
            vit(x, layer=22, feature_type="patch")
            torch.cuda.synchronize()
            t = ops.Timer(); t.start()
            for _ in range(2):
                vit(x, layer=22, feature_type="patch")
            t.stop(); ts[name].append(t.elapsed_ms() / 2)
    ops.set_option("gemm_stream_mb", -1)
    base = statistics.median(ts[POL[0][0]])
    print(f"B={B} @{res}: " + " | ".join(f"{n} {statistics.median(v):.3f} ms ({statistics.median(v) / base:.3f})" for n, v in ts.items()), flush=True)
