"""Single-node multi-GPU layer: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The hot path shards without a data-path exchange (proposals / frames / meshes are
independent: SURVEY §8e); the only collectives are tiny all-gathers of results:

  * bank-row sharding:  every rank scans its rows for all Q queries, all-gathers Q*k (score f32, index i32) pairs
    and merges them with the canonical (score desc, index asc) rule -> identical top-k on every rank;
  * proposal / frame / object sharding: ranks own disjoint work items and all-gather fixed-size result rows.

The reference has no counterpart (its parallelism is SLURM array jobs + files: scripts/extract_retrieval_features.py:32-34,
scripts/dino_inference.py:51-54, merge_results.py).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shared_gpu_allowed() -> bool:
    """FP_ALLOW_SHARED_GPU=1: several ranks may share one device (flow tests on a single-GPU box).  Without it, more ranks than
    visible GPUs is an error — a mis-provisioned box must not produce a plausible-looking N-"GPU" line."""
    return os.environ.get("FP_ALLOW_SHARED_GPU", "0") == "1"


def local_world(world: int) -> int:
    """ranks on THIS node: LOCAL_WORLD_SIZE when the launcher sets it (torch.distributed.run does), else the whole world — a two-node
    launch of 2 x 8 ranks must not be refused because 16 > 8 visible GPUs"""
    try:
        return int(os.environ.get("LOCAL_WORLD_SIZE", world))
    except ValueError:
        return world


def _require_own_devices(world: int) -> None:
    world = local_world(world)
    if torch.cuda.is_available() and world > torch.cuda.device_count() and not shared_gpu_allowed():
        raise SystemExit(f"freepose_amd: {world} ranks on this node but only {torch.cuda.device_count()} visible GPU(s); one process per GPU is the "
                         "contract.  Set FP_ALLOW_SHARED_GPU=1 to let ranks share a device (flow tests only: the result is stamped "
                         "shared_devices = true and n_gpus = the number of distinct devices).")


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank); initialises the default process group when launched by torch.distributed.run.  Refuses to
    put several ranks on one GPU unless FP_ALLOW_SHARED_GPU=1 (then the backend defaults to gloo: RCCL rejects shared devices)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _require_own_devices(world)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        chosen = backend is None
        if backend is None:   # RCCL ("nccl") on GPUs; FP_DIST_BACKEND=gloo lets several ranks share one GPU (single-GPU test boxes)
            backend = os.environ.get("FP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            if chosen and local_world(world) > torch.cuda.device_count():   # shared devices (allowed above): RCCL rejects them, the flow
                backend = os.environ.get("FP_DIST_BACKEND", "gloo")          # test runs on gloo — unless the caller named a backend
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if os.environ.get("FP_COMM_STACK", "torch") == "capi" and torch.cuda.is_available():
            use_capi_comm()
    elif torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    return rank, world, local


def rank_report() -> dict:
    """Who is running where: backend, RCCL version, and per rank the CUDA device index + PCI bus id, all-gathered so that rank 0
    can print it.  `devices_distinct` counts distinct (host-local) PCI bus ids; `shared_devices` is true when it is smaller than
    the world size — a line carrying it is a flow test, not a scaling measurement."""
    rank, ws = world()
    me = {"rank": rank, "device": None, "pci_bus_id": None, "name": None}
    if torch.cuda.is_available():
        d = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(d)
        bus = None
        if all(hasattr(pr, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        elif hasattr(pr, "uuid"):
            bus = str(pr.uuid)
        me.update(device=d, pci_bus_id=bus if bus is not None else f"index{d}", name=pr.name)
    ranks = [me]
    backend = "none"
    if ws > 1:
        backend = dist.get_backend()
        ranks = [None] * ws
        dist.all_gather_object(ranks, me)
    ids = {r["pci_bus_id"] for r in ranks if r["pci_bus_id"] is not None}
    distinct = len(ids) if ids else (1 if ws == 1 else 0)
    rccl = None
    try:
        v = torch.cuda.nccl.version()
        rccl = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        pass
    return {"backend": backend, "world_size": ws, "devices_distinct": distinct, "shared_devices": bool(ids) and distinct < ws,
            "rccl_version": rccl, "comm_stack": comm_stack(), "ranks": ranks}


def announce(tag: str = "dist") -> dict:
    """all ranks: gather rank_report(); rank 0 prints it as one `[tag] {json}` line on stderr (the CLIs call this once after
    init_from_env when world > 1, so a run's log says which backend and which devices produced it)"""
    import json
    import sys
    rep = rank_report()
    if world()[0] == 0:
        print(f"[{tag}] " + json.dumps(rep), file=sys.stderr, flush=True)
    return rep


def self_launch(n_ranks: int, target: Sequence[str], argv: Sequence[str]) -> None:
    """`python bench.py --gpus N` / `python -m scripts.dino_inference --gpus N` started WITHOUT a launcher: re-exec the same
    command as N ranks under torch.distributed.run on 127.0.0.1 (one process per GPU, RCCL) and exit with its status.  Returns
    immediately when n_ranks <= 1 or when this process already is a rank (WORLD_SIZE set by a launcher).  `target` is
    ["bench.py"] or ["-m", "scripts.dino_inference"].  With fewer visible GPUs than ranks the launch is REFUSED (exit status != 0)
    unless FP_ALLOW_SHARED_GPU=1 (single-GPU test boxes); then the ranks share devices, which RCCL rejects, so FP_DIST_BACKEND falls
    back to gloo unless the caller set it."""
    if n_ranks <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.is_available() and torch.cuda.device_count() < n_ranks:
        _require_own_devices(n_ranks)                       # exits non-zero unless FP_ALLOW_SHARED_GPU=1
        env.setdefault("FP_DIST_BACKEND", "gloo")           # RCCL refuses two ranks on one device
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), *target, *argv]
    raise SystemExit(subprocess.call(cmd, env=env))


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """contiguous, balanced [lo, hi) slice of n items (bank rows, meshes)"""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_items(n: int, rank: int, world_size: int) -> List[int]:
    """round-robin item ids (proposals, frames, objects): balances cost that grows with the index"""
    return list(range(rank, n, world_size))


def shard_chunk(n: int, rank: int, world_size: int) -> List[int]:
    """contiguous chunk of item ids (frames of one video when each rank tracks its own stretch of the clip)"""
    lo, hi = shard_range(n, rank, world_size)
    return list(range(lo, hi))


# ---- ONE data-path collective: an all-gather of equally-sized buffers ------------------------------------------------------------
# Every exchange of the path (top-k candidate pairs, pose rows, soft-vote lists) goes through _all_gather_equal().  Two transports
# sit behind it, chosen once per process:
#   * torch.distributed (default): RCCL through PyTorch's own copy ("nccl") on GPUs, gloo in the CPU tests / on shared devices;
#   * the library's C-ABI communicator (FP_COMM_STACK=capi, or use_capi_comm()): fp_comm_init + fp_allgather_bytes of
#     include/freepose_hip.h — what a host without torch binds; the process group is then only the bootstrap that
#     carries the 128-byte unique id from rank 0 to the others.
# With one rank both are the identity.
_capi = {"ctx": None}


def use_capi_comm() -> None:
    """route the data-path all-gathers of CUDA tensors through the library's communicator (RCCL opened by libfreepose_hip.so).  Needs an
    initialised process group (any backend) for the bootstrap when world > 1; one device per rank (RCCL's rule)."""
    if _capi["ctx"] is not None:
        return
    import ctypes as C
    from freepose_amd import _lib, ops
    lib = _lib.load()
    rank, ws = world()
    ctx = ops.context()
    uid = (C.c_ubyte * 128)()
    if rank == 0:
        _lib.check(lib.fp_comm_unique_id(uid), "fp_comm_unique_id")
    if ws > 1:
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0)
        uid = (C.c_ubyte * 128).from_buffer_copy(box[0])
    if lib.fp_comm_size(ctx) == 1 and lib.fp_comm_rank(ctx) == 0 and ws >= 1:
        rc = lib.fp_comm_init(ctx, ws, rank, uid)
        if rc != 0 and b"already" not in (lib.fp_last_error() or b""):
            _lib.check(rc, "fp_comm_init")
    _capi["ctx"] = ctx


def comm_stack() -> str:
    return "capi" if _capi["ctx"] is not None else "torch"


def _all_gather_equal(t: torch.Tensor) -> List[torch.Tensor]:
    """[t of rank 0, t of rank 1, ...] for a contiguous tensor of the same shape and dtype on every rank"""
    rank, ws = world()
    t = t.contiguous()
    if _capi["ctx"] is not None and t.is_cuda:
        from freepose_amd import _lib
        lib = _lib.load()
        n = int(lib.fp_comm_size(_capi["ctx"]))
        out = torch.empty((n,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        _lib.check(lib.fp_allgather_bytes(_capi["ctx"], _lib.ptr(t), t.numel() * t.element_size(), _lib.ptr(out), _lib.current_stream()),
                   "fp_allgather_bytes")
        return list(out.unbind(0))
    if ws == 1:
        return [t]
    parts = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(parts, t)
    return parts


def all_gather_cat(t: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """all-gather equally-shaped tensors and concatenate along `dim` (rank order)."""
    parts = _all_gather_equal(t)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=dim)


def all_gather_rows(rows: torch.Tensor, counts: Sequence[int] | None = None) -> torch.Tensor:
    """all-gather a variable number of fixed-width rows per rank (pads to the max count)."""
    rank, ws = world()
    if ws == 1:
        return rows
    n = torch.tensor([rows.shape[0]], device=rows.device, dtype=torch.int64)
    ns = [int(x.item()) for x in _all_gather_equal(n)]
    mx = max(ns)
    pad = torch.zeros((mx,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    parts = _all_gather_equal(pad)
    return torch.cat([p[:k] for p, k in zip(parts, ns)], dim=0)


def merge_topk(cand_scores: torch.Tensor, cand_idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """canonical (score desc, index asc) merge of candidate lists [Q, C] -> [Q, k].  GPU tensors go through the HIP
    kernel (fp_topk_merge); CPU tensors (gloo tests) through a stable numpy sort with the same ordering."""
    if cand_scores.is_cuda:
        from freepose_amd import ops
        return ops.topk_merge(cand_scores, cand_idx, k)
    s, i = cand_scores.numpy(), cand_idx.numpy()
    order = np.lexsort((i, -s), axis=1)[:, :k]
    return torch.from_numpy(np.take_along_axis(s, order, 1)), torch.from_numpy(np.take_along_axis(i, order, 1))


PAD_SCORE = float("-inf")          # sentinel candidates of a shard with fewer than k rows: sort behind every real score ...
PAD_INDEX = 2 ** 31 - 1            # ... and behind every real index


def pad_candidates(s: torch.Tensor, i: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """pad per-rank candidate lists [Q, kk <= k] to [Q, k] so that every rank contributes equally-shaped tensors to the
    all-gather (shards differ in size by one row; with k > rows-per-shard the unpadded shapes would differ and hang RCCL)"""
    kk = s.shape[1]
    if kk == k:
        return s, i
    ps = torch.full((s.shape[0], k), PAD_SCORE, dtype=s.dtype, device=s.device)
    pi = torch.full((i.shape[0], k), PAD_INDEX, dtype=i.dtype, device=i.device)
    ps[:, :kk], pi[:, :kk] = s, i
    return ps, pi


def sharded_bank_topk(local_topk_fn, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """bank-row sharding: `local_topk_fn(queries, k)` returns this rank's (scores, GLOBAL indices), at most k and padded
    here to exactly k; one all-gather of Q*k pairs, then the same merge on every rank.  k must not exceed the bank size."""
    s, i = pad_candidates(*local_topk_fn(queries, k), k)
    return merge_topk(all_gather_cat(s, dim=1), all_gather_cat(i, dim=1), k)


def soft_vote_reduce(scores: torch.Tensor, idx: torch.Tensor, frame_ids: torch.Tensor, n_rows: int):
    """Video soft-vote as a sharded reduction (scripts/extract_proposals_ground_video.py:154-159,186-190: per frame and
    object a dense [N] vector that is zero except at the frame's top-k rows, mean over frames, top-1 per object).

    scores f32 / idx i32 [F_local, n_obj, k]: the sparse per-frame lists of THIS rank's frames; frame_ids i64 [F_local]
    their global frame numbers.  The sparse lists are all-gathered (k*8 bytes per frame-object), and every rank builds the
    same dense mean: frames are accumulated in ascending frame order with one float32 add per (frame, row) — a fixed order,
    so all ranks (and a single-rank run) get bit-identical means; ties go to the lowest row.  Returns
    (best_row i64 [n_obj], best_score f32 [n_obj], mean f32 [n_obj, n_rows])."""
    F_loc, n_obj, k = scores.shape
    s = all_gather_rows(scores.reshape(F_loc, n_obj * k).contiguous())
    i = all_gather_rows(idx.reshape(F_loc, n_obj * k).contiguous())
    f = all_gather_rows(frame_ids.reshape(F_loc, 1).to(scores.device))[:, 0]
    order = torch.argsort(f, stable=True)
    s, i = s[order].reshape(-1, n_obj, k), i[order].reshape(-1, n_obj, k).long()
    acc = torch.zeros((n_obj, n_rows), dtype=torch.float32, device=scores.device)
    for fr in range(s.shape[0]):
        acc.scatter_add_(1, i[fr], s[fr])            # a frame's top-k rows are distinct: no colliding adds
    acc /= float(s.shape[0])
    best = acc.max(dim=1).values
    rows = torch.arange(n_rows, device=acc.device)[None].expand(n_obj, -1)
    best_row = torch.where(acc == best[:, None], rows, torch.full_like(rows, n_rows)).min(dim=1).values
    return best_row, best, acc
