"""Multi-GPU plumbing on real devices.  (1) the C-ABI communication entry points (fp_comm_*, RCCL opened lazily) with a
single-rank communicator on the one GPU every box has; (2) when the box has >= 2 GPUs: the torch.distributed path over the
`nccl` backend (= RCCL over xGMI) — sharded bank top-k, variable-row all-gather of float64 CUDA rows, the soft-vote reduction —
and the C-ABI all-gathers across ranks.  (2) is skipped on single-GPU boxes; the same code runs there over gloo
(tests/test_gpu_multirank.py, tests/test_distributed_cpu.py)."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_c_abi_comm_single_rank():
    from freepose_amd import _lib, ops
    lib = _lib.load()
    ctx = ops.context()
    assert lib.fp_comm_size(ctx) == 1 and lib.fp_comm_rank(ctx) == 0
    rng = np.random.default_rng(4)
    s = torch.from_numpy(np.sort(rng.random((3, 16)).astype(np.float32), axis=1)[:, ::-1].copy()).cuda()
    i = torch.from_numpy(rng.permutation(48).reshape(3, 16).astype(np.int32)).cuda()
    os_, oi = torch.empty((3, 8), device="cuda"), torch.empty((3, 8), dtype=torch.int32, device="cuda")

    def gather_topk():
        _lib.check(lib.fp_allgather_topk(ctx, _lib.ptr(s), _lib.ptr(i), 3, 16, 8, _lib.ptr(os_), _lib.ptr(oi), _lib.current_stream()),
                   "fp_allgather_topk")
        torch.cuda.synchronize()
        ms, mi = ops.topk_merge(s, i, 8)
        assert torch.equal(os_, ms) and torch.equal(oi, mi)
    gather_topk()                                   # no communicator: the gather is a copy
    uid = (C.c_char * 128)()
    _lib.check(lib.fp_comm_unique_id(uid), "fp_comm_unique_id")
    _lib.check(lib.fp_comm_init(ctx, 1, 0, uid), "fp_comm_init")
    try:
        assert lib.fp_comm_size(ctx) == 1 and lib.fp_comm_rank(ctx) == 0
        gather_topk()                               # through ncclAllGather on a 1-rank communicator
        rows = torch.arange(38, dtype=torch.float64, device="cuda").reshape(2, 19)
        out = torch.zeros_like(rows)
        _lib.check(lib.fp_allgather_poses(ctx, _lib.ptr(rows), 2, 19, _lib.ptr(out), _lib.current_stream()), "fp_allgather_poses")
        torch.cuda.synchronize()
        assert torch.equal(out, rows)
        assert lib.fp_comm_init(ctx, 1, 0, uid) != 0 and b"already" in lib.fp_last_error()
    finally:
        _lib.check(lib.fp_comm_destroy(ctx), "fp_comm_destroy")
    assert lib.fp_comm_size(ctx) == 1


def test_one_comm_stack_single_rank():
    """parallel.* has ONE data-path collective (_all_gather_equal) with two transports; with the library's communicator selected
    (FP_COMM_STACK=capi / use_capi_comm) the drivers' calls go through fp_allgather_bytes — on one rank the identity, like the
    torch.distributed transport.  Runs in a subprocess: the choice is per process."""
    code = (
        "import torch, numpy as np\n"
        "from freepose_amd import _lib, ops, parallel\n"
        "import bench\n"
        "from freepose_amd.retrieval import TemplateBank\n"
        "lib = _lib.load(); calls = []\n"
        "assert parallel.comm_stack() == 'torch'\n"
        "rows = torch.arange(38, dtype=torch.float64, device='cuda').reshape(2, 19)\n"
        "a = parallel.all_gather_rows(rows); b = parallel.all_gather_cat(rows, dim=1)\n"
        "parallel.use_capi_comm(); parallel.use_capi_comm()\n"
        "assert parallel.comm_stack() == 'capi' and lib.fp_comm_size(ops.context()) == 1\n"
        "orig = lib.fp_allgather_bytes\n"
        "assert torch.equal(parallel.all_gather_rows(rows), a) and torch.equal(parallel.all_gather_cat(rows, dim=1), b)\n"
        "parts = parallel._all_gather_equal(rows)\n"
        "assert len(parts) == 1 and torch.equal(parts[0], rows) and parts[0].data_ptr() != rows.data_ptr()\n"
        "bank = TemplateBank(bench.synthetic_bank(5000, 1024, seed=3))\n"
        "q = ops.l2_normalize(torch.randn((3, 1024), device='cuda').to(torch.bfloat16))\n"
        "s, i = parallel.sharded_bank_topk(bank._local_topk, q, 100); s0, i0 = bank.topk(q, 100)\n"
        "assert torch.equal(s, s0) and torch.equal(i, i0)\n"
        "r, bst = bank.soft_vote([q, q], k=50)\n"
        "assert parallel.rank_report()['comm_stack'] == 'capi'\n"
        "print('ONE_STACK_OK')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=str(ROOT)))
    assert r.returncode == 0 and "ONE_STACK_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_rccl_two_ranks(tmp_path):
    """2 ranks over the `nccl` backend (RCCL) + the C-ABI communicator.  With >= 2 GPUs each rank has its own device; on a
    single-GPU box both ranks are pointed at GPU 0 — RCCL normally refuses that ("Duplicate GPU detected"), in which case the
    test is skipped with RCCL's own message; any other failure is a failure."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FP_DIST_BACKEND="nccl", FP_ALLOW_SHARED_GPU="1", FP_COMM_DIR=str(tmp_path), NCCL_DEBUG="WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29581", str(ROOT / "tests" / "_multirank_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and torch.cuda.device_count() < 2:
        blob = r.stdout + r.stderr
        refused = [ln.strip() for ln in blob.splitlines() if "Duplicate GPU" in ln or "invalid usage" in ln or "ncclInvalidUsage" in ln]
        if refused:
            print("RCCL refused 2 ranks on one device:", refused[0])
            pytest.skip(f"single-GPU box and RCCL refuses two ranks on one device: {refused[0][:200]}")
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "MULTIRANK_BANK_OK 2" in r.stdout and "MULTIRANK_CABI_OK 2" in r.stdout and "MULTIRANK_ONE_STACK_OK 2" in r.stdout


def test_comm_init_fails_cleanly_when_the_ranks_never_all_arrive():
    """the first 8-GPU lease must not hang on a launcher mistake: fp_comm_init is a rendezvous of `nranks` ranks —
    ranks that disagree on nranks (here: rank 0 believes in 2 ranks, nobody else exists) would wait for ever inside ncclCommInitRank.
    With "comm_timeout_s" the call returns FP_ERR_STATE + a message instead; a blank id and a bad rank are refused at once; the
    context keeps working without a communicator.  Runs in its own process (the abandoned rendezvous thread stays inside RCCL)."""
    code = r"""
import ctypes as C, sys, time, os
import torch
from freepose_amd import _lib, ops
lib = _lib.load()
ctx = ops.context()
blank = (C.c_char * 128)()
assert lib.fp_comm_init(ctx, 2, 0, blank) != 0 and b"all zero" in lib.fp_last_error()
uid = (C.c_char * 128)()
_lib.check(lib.fp_comm_unique_id(uid), "uid")
assert lib.fp_comm_init(ctx, 2, 2, uid) != 0 and b"bad argument" in lib.fp_last_error()
_lib.check(lib.fp_ctx_set_option(ctx, b"comm_timeout_s", 3), "opt")
t = time.time()
rc = lib.fp_comm_init(ctx, 2, 0, uid)
dt = time.time() - t
msg = lib.fp_last_error().decode()
assert rc == 3 and "rendezvous of 2 ranks" in msg and "same nranks" in msg, (rc, msg)
assert 2.5 <= dt <= 30, dt
assert lib.fp_comm_size(ctx) == 1 and lib.fp_comm_rank(ctx) == 0
x = torch.arange(64, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
_lib.check(lib.fp_allgather_bytes(ctx, _lib.ptr(x), 256, _lib.ptr(y), _lib.current_stream()), "copy")
torch.cuda.synchronize()
assert torch.equal(x, y)
print("COMM_TIMEOUT_OK %.1f" % dt, flush=True)
os._exit(0)
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "COMM_TIMEOUT_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-2500:])
