"""Upper bound of merging the qk and V projections into one N = 3072 launch: LN-folded plain GEMM at N = 3072 (V row-major) against
the shipped pair (N = 2048 plain + N = 1024 with the transposed per-head store), same process, alternating.  Development probe."""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()   # measurement variants / hooks live in libfreepose_hip_lab.so only (python -m freepose_amd.build --lab)
from freepose_amd import ops  # noqa: E402

B, npad, K = 214, 1376, 1024
M = B * npad
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
g_ln = (1 + 0.1 * torch.randn(K, device="cuda")).to(torch.bfloat16)
b_ln = (0.1 * torch.randn(K, device="cuda")).to(torch.bfloat16)
w3 = (torch.randn(3072, K, device="cuda") * 0.02).to(torch.bfloat16)
b3 = torch.zeros(3072, device="cuda").to(torch.bfloat16)
w2, bq, wv, bv = w3[:2048].contiguous(), b3[:2048].contiguous(), w3[2048:].contiguous(), b3[2048:].contiguous()


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    tm = ops.Timer(); tm.start()
    for _ in range(n):
        fn()
    tm.stop()
    return tm.elapsed_ms() / n


res = {"merged": [], "pair": []}
for _ in range(5):
    res["merged"].append(t(lambda: ops.ln_linear(x, g_ln, b_ln, w3, b3, mode=0)))
    res["pair"].append(t(lambda: (ops.ln_linear(x, g_ln, b_ln, w2, bq, mode=0), ops.ln_linear(x, g_ln, b_ln, wv, bv, mode=2, npad=npad, heads=16))))
for k, v in res.items():
    print(f"{k}: median {statistics.median(v):.3f} ms (min {min(v):.3f})")
w64, b64 = w3[:64].contiguous(), b3[:64].contiguous()
ov = statistics.median([t(lambda: ops.ln_linear(x, g_ln, b_ln, w64, b64, mode=0)) for _ in range(5)])
print(f"per-call overhead of the test op (row statistics + weight fold + an N = 64 GEMM): {ov:.3f} ms -> merged GEMM ~{statistics.median(res['merged']) - ov:.3f} ms, "
      f"pair of GEMMs ~{statistics.median(res['pair']) - 2 * ov:.3f} ms")
