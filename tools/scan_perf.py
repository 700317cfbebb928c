"""Bank scan / top-k timing probe (development).  Run under `rocprofv3 --kernel-trace --stats` for per-kernel durations."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops  # noqa: E402
from freepose_amd.retrieval import TemplateBank  # noqa: E402


def main():
    N, D = 46037, 1024
    rng = np.random.default_rng(0)
    bank = rng.standard_normal((N, D)).astype(np.float32)
    tb = TemplateBank(bank, shard=False)
    for Q in (1, 4, 16):
        q = ops.l2_normalize(torch.from_numpy(rng.standard_normal((Q, D)).astype(np.float32)).cuda().to(torch.bfloat16))
        tb.topk(q, 100)
        torch.cuda.synchronize()
        t = ops.Timer()
        t.start()
        for _ in range(50):
            tb.topk(q, 100)
        t.stop()
        ms = t.elapsed_ms() / 50
        print(f"Q={Q}: scan+select {ms * 1e3:.1f} us per call  ({N * D * 2 / ms / 1e6:.0f} GB/s per pass incl. select)", flush=True)


if __name__ == "__main__":
    main()
