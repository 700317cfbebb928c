"""Template scorer timing: the reference's cached path scores T = 600 templates x P = 900 patches x D = 1024 per
proposal (pose_estimator.py:85-90).  Raw features (normalised on the fly, fp_template_score) vs the pre-normalised store
(fp_template_score_normed) vs the one-off in-place normalisation; HIP events, medians; 1.106 GB per pass.  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel rows.  python tools/template_score_perf.py"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops  # noqa: E402


def med(fn, n=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    t = ops.Timer()
    for _ in range(n):
        t.start()
        fn()
        t.stop()
        ts.append(t.elapsed_ms())
    return statistics.median(ts)


def main():
    for (T, P, D) in ((600, 900, 1024), (576, 1369, 1024)):
        raw = torch.randn(T, P, D, device="cuda").to(torch.bfloat16)
        q = ops.l2_normalize(torch.randn(P, D, device="cuda").to(torch.bfloat16))
        nbytes = T * P * D * 2
        t_raw = med(lambda: ops.template_score(raw, q))
        tn = raw.clone()
        t_norm = med(lambda: ops.l2_normalize(tn, inplace=True), n=5)
        tn = ops.l2_normalize(raw.clone(), inplace=True)
        t_pre = med(lambda: ops.template_score(tn, q, normalized=True))
        same = torch.equal(ops.template_score(raw, q), ops.template_score(tn, q, normalized=True))
        print(f"T={T} P={P} D={D} ({nbytes / 1e9:.3f} GB): raw features {t_raw:.3f} ms = {nbytes / t_raw / 1e9:.2f} TB/s | pre-normalised store "
              f"{t_pre:.3f} ms = {nbytes / t_pre / 1e9:.2f} TB/s ({nbytes / t_pre / 1e9 / 8 * 100:.0f} % of 8 TB/s) | one-off in-place normalisation "
              f"{t_norm:.3f} ms | identical scores: {same}", flush=True)


if __name__ == "__main__":
    main()
