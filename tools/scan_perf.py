"""Bank scan / top-k timing probe (development): WARM (the 94 MB bank stays in the 256 MiB Infinity Cache between back-to-back
scans) and COLD (a 1 GiB buffer is rewritten between scans, as the 1.6 GB of ViT activations do in the pipeline) are reported
separately; per-kernel times come from HIP events around single calls.  Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel durations (`bank_scan_kernel` rows: the cold calls are the ones behind a fill kernel)."""
import statistics
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops  # noqa: E402
from freepose_amd.retrieval import TemplateBank  # noqa: E402


import os
EVICT = os.environ.get("SCAN_EVICT", "read")


def main():
    N, D = 46037, 1024
    print("eviction sweep:", EVICT)
    rng = np.random.default_rng(0)
    bank = rng.standard_normal((N, D)).astype(np.float32)
    tb = TemplateBank(bank, shard=False)
    evict = torch.empty(1 << 28, dtype=torch.float32, device="cuda")     # 1 GiB > L2 + MALL
    nbytes = N * D * 2
    for Q in (1, 4, 16):
        q = ops.l2_normalize(torch.from_numpy(rng.standard_normal((Q, D)).astype(np.float32)).cuda().to(torch.bfloat16))
        tb.topk(q, 100)
        torch.cuda.synchronize()
        t = ops.Timer()
        t.start()
        for _ in range(50):
            tb.topk(q, 100)
        t.stop()
        warm = t.elapsed_ms() / 50
        cold = []
        for i in range(12):
            if EVICT == "write":
                evict.fill_(float(i))                                     # evicts the bank; leaves 1 GiB of DIRTY lines behind
            else:
                sink = evict.view(torch.int32).max()                      # evicts the bank with clean lines (read-only sweep)
            torch.cuda.synchronize()
            t.start()
            tb.topk(q, 100)
            t.stop()
            cold.append(t.elapsed_ms())
        c = statistics.median(cold)
        print(f"Q={Q}: scan+select  warm (MALL-resident bank) {warm * 1e3:.1f} us/call = {nbytes / warm / 1e6:.0f} GB/s;  "
              f"cold (bank evicted by a 1 GiB {EVICT} sweep before each call) {c * 1e3:.1f} us/call = {nbytes / c / 1e6:.0f} GB/s  (stage incl. select + merge)", flush=True)


if __name__ == "__main__":
    main()
