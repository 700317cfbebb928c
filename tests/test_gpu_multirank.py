"""The driver's multi-GPU launch shape (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) on a
single-GPU box: two ranks share the device through the gloo backend (FP_DIST_BACKEND), which exercises the whole rank flow —
process-group init from the environment, proposal sharding, the all-gather of result rows, barriers, the MAX-reduced
timing and rank-0 reporting.  Throughput is meaningless here; the 8-GPU numbers come from the driver's RCCL run."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _n_gpus():
    import torch
    return torch.cuda.device_count()


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, FP_DIST_BACKEND="gloo", FP_ALLOW_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--hyp", "24",
           "--bank", "2000", "--mesh-sub", "3", "--vit-batch", "8"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    out = json.loads(lines[0])
    distinct = min(2, _n_gpus())
    assert out["n_ranks"] == 2 and out["n_gpus"] == distinct and out["shared_devices"] == (distinct < 2)
    assert out["backend"] == "gloo" and len(out["ranks"]) == 2 and {r["rank"] for r in out["ranks"]} == {0, 1}
    assert all(r["pci_bus_id"] for r in out["ranks"]) and len(out["ms_per_step_ranks"]["all"]) == 2
    assert out["ms_per_step_ranks"]["max"] == pytest.approx(out["ms_per_step"], rel=1e-6)
    assert out["scaling"] == "weak" and out["cpu_baseline"] is None
    assert out["value"] > 0 and out["config"]["proposals_per_step_per_gpu"] == 1
    assert out["roofline"]["bound"] == "mfma" and out["roofline"]["achieved"] > 0
    # the N-rank line carries BOTH video legs: the deviating frame-chunk one and the exact object-sharded one (strong scaling)
    vw = out["video_workload"]
    assert "frame chunks" in vw["sharding"] and vw["object_sharded"]["scaling"] == "strong"
    assert vw["object_sharded"]["objects_per_rank"] == [4, 4] and vw["object_sharded"]["value"] > 0
    assert out.get("comm_stack", "torch") == "torch"


def test_sharded_bank_topk_two_ranks_on_one_gpu():
    """bank-row sharding with the candidate all-gather and the HIP merge kernel == unsharded scan, on every rank"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FP_ALLOW_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29562", str(ROOT / "tests" / "_multirank_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "MULTIRANK_BANK_OK 2" in r.stdout


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with NO launcher (how the round driver may start it): bench.py re-execs itself as 2 ranks under
    torch.distributed.run (RCCL when the box has >= 2 GPUs).  On a box with FEWER GPUs than ranks the launch is refused with a
    non-zero status — a mis-provisioned box must not print a plausible 2-"GPU" line — unless FP_ALLOW_SHARED_GPU=1, and then the
    line is stamped shared_devices = true, n_gpus = the number of distinct devices."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FP_DIST_BACKEND", "FP_ALLOW_SHARED_GPU")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--hyp", "24", "--bank", "2000",
           "--mesh-sub", "3", "--vit-batch", "8", "--video-frames", "0"]
    shared = _n_gpus() < 2
    if shared:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "FP_ALLOW_SHARED_GPU" in (r.stderr + r.stdout), (r.returncode, r.stderr[-1500:])
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln], "a refused launch prints no result line"
        # the same refusal when a launcher starts the ranks itself (the driver's launch shape)
        lcmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29563"] + cmd[1:]
        r = subprocess.run(lcmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        env["FP_ALLOW_SHARED_GPU"] = "1"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_ranks"] == 2 and out["value"] > 0 and out["steps"] == 1
    assert out["shared_devices"] == shared and out["n_gpus"] == out["devices_distinct"] == (1 if shared else 2)
    assert out["backend"] == ("gloo" if shared else "nccl") and out["rccl_version"]


def test_bench_eight_ranks_on_one_gpu():
    """the driver's N = 8 launch shape, end to end, before an 8-GPU node ever sees it: eight ranks share the one
    GPU over gloo and run the whole bench flow — rendezvous on 127.0.0.1, proposal sharding, result all-gather, barriers, MAX-reduced
    timing, one JSON line from rank 0 — and BOTH video legs (16 frames chunked over 8 ranks; 8 objects dealt one per rank).  Ranks with
    an EMPTY shard are the worker's case below (5 frames over 8 ranks)."""
    env = dict(os.environ, FP_DIST_BACKEND="gloo", FP_ALLOW_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29568", str(ROOT / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--hyp", "24",
           "--bank", "2000", "--mesh-sub", "3", "--vit-batch", "8", "--video-frames", "16"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    out = json.loads(lines[0])
    assert out["n_ranks"] == 8 and out["shared_devices"] == (_n_gpus() < 8) and len(out["ranks"]) == 8
    assert {r["rank"] for r in out["ranks"]} == set(range(8)) and len(out["ms_per_step_ranks"]["all"]) == 8
    assert out["value"] > 0 and out["scaling"] == "weak" and out["config"]["proposals_per_step_per_gpu"] == 1
    vw = out["video_workload"]
    assert "frame chunks" in vw["sharding"] and vw["value"] > 0
    assert vw["object_sharded"]["objects_per_rank"] == [1] * 8 and vw["object_sharded"]["value"] > 0


def test_sharded_bank_topk_eight_ranks_on_one_gpu():
    """bank-row sharding, variable-row gathers and the frame-sharded soft vote (5 frames over 8 ranks: three ranks hold no frame) at
    the node's rank count"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FP_ALLOW_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29569", str(ROOT / "tests" / "_multirank_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "MULTIRANK_BANK_OK 8" in r.stdout
