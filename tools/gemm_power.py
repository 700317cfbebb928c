"""Shader clock and socket power while the ViT GEMM shapes run (development probe, round 3): each shape loops for ~2.5 s while a
thread polls `rocm-smi -P -c --json`; reported per shape: TFLOP/s (HIP events), median socket power, median sclk.  Operands: uniform
random bf16 (the bench's regime) and all-zero (the same instruction stream with no data toggling).  FP_GEMM_DBG=8 in the environment
runs the persistent kernels WITHOUT their epilogues (main loop only).  python tools/gemm_power.py"""
import json
import os
import statistics
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import _lib  # noqa: E402
_lib.use_lab()   # measurement variants / hooks live in libfreepose_hip_lab.so only (python -m freepose_amd.build --lab)
from freepose_amd import ops  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop_flag = False
        self.power, self.sclk = [], []

    def run(self):
        while not self.stop_flag:
            try:
                out = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=5).stdout
                d = json.loads(out)
                card = d[sorted(d)[0]]
                for k, v in card.items():
                    kl = k.lower()
                    if "power" in kl and "(w)" in kl:
                        self.power.append(float(v))
                    if kl.startswith("sclk") and "mhz" in str(v).lower():
                        self.sclk.append(float(str(v).lower().replace("(", "").replace(")", "").replace("mhz", "").strip()))
            except Exception:
                pass
            time.sleep(0.05)


def run_shape(name, fn, flops, seconds=2.5):
    fn()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    t0 = time.perf_counter()
    tm = ops.Timer()
    n = 0
    tm.start()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    tm.stop()
    ms = tm.elapsed_ms() / n
    s.stop_flag = True
    s.join()
    pw = statistics.median(s.power) if s.power else float("nan")
    ck = statistics.median(s.sclk) if s.sclk else float("nan")
    print(f"{name:44s} {flops / ms / 1e9:7.0f} TF   power {pw:6.0f} W (max {max(s.power) if s.power else float('nan'):.0f})   sclk {ck:5.0f} MHz   "
          f"({len(s.power)} samples)", flush=True)


def main():
    M = 214 * 1376
    print("FP_GEMM_DBG =", os.environ.get("FP_GEMM_DBG", "0"), " M =", M)
    for fill in (("random",) if os.environ.get("FP_GEMM_DBG", "0") not in ("0", "8") else ("random", "zero")):
        for (N, K, epi, label) in [(2048, 1024, 0, "qk"), (1024, 1024, 2, "proj"), (4096, 1024, 1, "fc1"), (1024, 4096, 2, "fc2")]:
            if fill == "random":
                x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
                w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
                r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
            else:
                x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
                w = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
                r = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            b = torch.zeros(N, device="cuda").to(torch.bfloat16)
            g = torch.ones(N, device="cuda").to(torch.bfloat16)
            o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            run_shape(f"{label} N={N} K={K} epi={epi} {fill}", lambda: ops.gemm(x, w, b, epi, gamma=g, resid=r, out=o), 2.0 * M * N * K)
            del x, w, r, o
    # idle reference
    s = Sampler()
    s.start()
    time.sleep(1.0)
    s.stop_flag = True
    s.join()
    if s.power:
        print(f"idle: power {statistics.median(s.power):.0f} W, sclk {statistics.median(s.sclk) if s.sclk else float('nan'):.0f} MHz")


if __name__ == "__main__":
    main()
