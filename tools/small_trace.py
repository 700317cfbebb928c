"""Per-kernel timeline of small-batch ViT-L forwards (the launch sizes of BASELINE configs 3 and 5).

    python tools/small_trace.py run B res [n]        # n forwards of B crops @res^2 (what rocprofv3 --kernel-trace wraps)
    python tools/small_trace.py read <trace.csv> n   # per-kernel average, launches per forward, busy / idle time per forward

The reader takes the LAST n forwards of the trace (warm-up dropped by a marker gap: forwards are separated by a host sync)."""
import csv
import collections
import re
import sys
from pathlib import Path


def run(B, res, n):
    import torch
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from freepose_amd import ops
    vit = ops.ViT("dinov2_vitl14_reg", seed=0)
    x = torch.rand((B, 3, res, res), device="cuda").to(torch.bfloat16)
    for _ in range(3):
        vit(x, layer=22, feature_type="patch")
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t = ops.Timer(); t.start()
        vit(x, layer=22, feature_type="patch")
        t.stop(); ts.append(t.elapsed_ms())
        torch.cuda.synchronize()
    ts.sort()
    print(f"B={B} @{res}: median {ts[len(ts) // 2]:.3f} ms per forward ({vit.flops(B, res, res, 22) / ts[len(ts) // 2] / 1e9:.0f} TF)", flush=True)


def short(k):
    k = k.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^>]*>)?", k)
    return (m.group(1) + (m.group(2) or "")) if m else k[:60]


def read(path, n):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # forwards end with layernorm_kernel (final norm + slice)
    ends = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("layernorm_kernel")]
    assert len(ends) >= n + 1, (len(ends), n)
    lo = ends[-n - 1] + 1
    sel = rows[lo: ends[-1] + 1]
    per = collections.defaultdict(lambda: [0.0, 0])
    busy = 0.0
    idle = 0.0
    prev_end = None
    fwd_first = True
    wall = 0.0
    start = None
    for i, r in enumerate(sel):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = short(r["Kernel_Name"])
        per[k][0] += e - s
        per[k][1] += 1
        busy += e - s
        if start is None:
            start = s
        if prev_end is not None and not fwd_first:
            idle += max(0, s - prev_end)
        fwd_first = False
        prev_end = e
        if k.startswith("layernorm_kernel"):
            wall += e - start
            start = None
            fwd_first = True
    print(f"{path}: {n} forwards, {len(sel) / n:.1f} launches per forward")
    print(f"  per forward: first-start..last-end {wall / n / 1e3:.1f} us, kernels busy {busy / n / 1e3:.1f} us, gaps between kernels {idle / n / 1e3:.1f} us")
    for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:58s} calls/fwd={c / n:6.1f} avg_us={t / c / 1e3:8.2f} us/fwd={t / n / 1e3:8.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 10)
    else:
        read(sys.argv[2], int(sys.argv[3]))
