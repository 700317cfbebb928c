"""Checker-side helper (test infrastructure, like oracle/): the oracle ViT (oracle/vit_ref.py, torch on the CPU) over many crops, spread
over worker PROCESSES, each pinned to its own 16 logical CPUs.  One torch process does not scale past ~16 threads on the pool's 256-thread
hosts (bench.py's cpu_baseline probe), but disjoint 16-thread processes do, so the 576-hypothesis / 518^2 parity case costs the GPU box
about a minute of wall clock instead of four.  Nothing from freepose_amd is imported here; state dict and crops travel as files."""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
THREADS = 16


def _worker(argv):
    sd_path, crops_path, out_path, layer, dtype_name, batch, cpus = argv
    cpus = [int(c) for c in cpus.split(",") if c]
    if cpus and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    torch.set_num_threads(min(THREADS, len(cpus)) if cpus else THREADS)
    sys.path.insert(0, str(ROOT))
    from oracle import vit_ref
    dtype = getattr(torch, dtype_name)
    sd = torch.load(sd_path, map_location="cpu")
    crops = torch.load(crops_path, map_location="cpu")
    out = []
    with torch.inference_mode():
        for i in range(0, crops.shape[0], int(batch)):
            out.append(vit_ref.vit_forward(sd, crops[i:i + int(batch)], layer=int(layer), feature_type="patch", dtype=dtype).to(torch.bfloat16))
    torch.save(torch.cat(out), out_path)


def n_workers(n_crops: int) -> int:
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(8, avail // THREADS, n_crops // 32))


def oracle_feats(sd, crops_f32, layer, dtype, batch=8):
    """bf16 patch features [B,P,D] of `crops_f32` from the oracle ViT computing in `dtype` (state dict `sd` already in that dtype)"""
    w = n_workers(crops_f32.shape[0])
    if w == 1:
        from oracle import vit_ref
        nthr = torch.get_num_threads()
        torch.set_num_threads(min(THREADS, nthr))
        try:
            out = []
            with torch.inference_mode():
                for i in range(0, crops_f32.shape[0], batch):
                    out.append(vit_ref.vit_forward(sd, crops_f32[i:i + batch], layer=layer, feature_type="patch", dtype=dtype).to(torch.bfloat16))
        finally:
            torch.set_num_threads(nthr)
        return torch.cat(out)
    cpus = sorted(os.sched_getaffinity(0))
    per = -(-crops_f32.shape[0] // w)
    per = -(-per // batch) * batch
    with tempfile.TemporaryDirectory(prefix="fp_oracle_pool_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        sd_path = os.path.join(tmp, "sd.pt")
        torch.save(sd, sd_path)
        procs, outs = [], []
        for k in range(w):
            part = crops_f32[k * per:(k + 1) * per]
            if part.shape[0] == 0:
                continue
            cp, op = os.path.join(tmp, f"crops{k}.pt"), os.path.join(tmp, f"out{k}.pt")
            torch.save(part.clone(), cp)
            mine = ",".join(str(c) for c in cpus[k * THREADS:(k + 1) * THREADS])
            env = dict(os.environ, OMP_NUM_THREADS=str(THREADS), MKL_NUM_THREADS=str(THREADS), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
            procs.append(subprocess.Popen([sys.executable, __file__, sd_path, cp, op, str(layer), str(dtype).split(".")[-1], str(batch), mine], env=env))
            outs.append(op)
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError("an oracle ViT worker failed")
        return torch.cat([torch.load(o, map_location="cpu") for o in outs])


if __name__ == "__main__":
    _worker(sys.argv[1:8])
