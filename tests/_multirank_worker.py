"""Worker for tests/test_gpu_multirank.py (launched by torch.distributed.run, several ranks sharing one GPU over gloo):
bank-row sharding — every rank scans its rows, candidates are all-gathered and merged with the canonical rule — must
reproduce the unsharded top-k bit for bit on every rank; frame-sharded soft voting must agree across ranks."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from freepose_amd import ops, parallel  # noqa: E402
from freepose_amd.retrieval import TemplateBank  # noqa: E402


def _soft_vote_local(votes, N):
    """single-process restatement of the reduction (frame order, one float32 add per frame and row)"""
    n_obj = votes[0][0].shape[0]
    acc = torch.zeros((n_obj, N), dtype=torch.float32, device="cuda")
    for s, i in votes:
        acc.scatter_add_(1, i.long(), s)
    acc /= float(len(votes))
    best = acc.max(dim=1).values
    rows = torch.arange(N, device="cuda")[None].expand(n_obj, -1)
    return torch.where(acc == best[:, None], rows, torch.full_like(rows, N)).min(dim=1).values.cpu().numpy(), best.cpu().numpy(), acc


def _c_abi_comm(rank, world, comm_dir, shape, sharded, qd, s_full, i_full):
    """the C-ABI communicator across ranks: unique id from rank 0 through a file, then fp_allgather_topk on the local lists"""
    import ctypes as C
    import time
    from freepose_amd import _lib
    lib = _lib.load()
    ctx = ops.context()
    uid = (C.c_char * 128)()
    f = comm_dir / "rccl_uid.bin"
    if rank == 0:
        _lib.check(lib.fp_comm_unique_id(uid), "fp_comm_unique_id")
        tmp = comm_dir / "rccl_uid.tmp"
        tmp.write_bytes(bytes(uid))
        tmp.rename(f)
    else:
        for _ in range(600):
            if f.exists():
                break
            time.sleep(0.05)
        C.memmove(uid, f.read_bytes(), 128)
    _lib.check(lib.fp_comm_init(ctx, world, rank, uid), "fp_comm_init")
    s_loc, i_loc = parallel.pad_candidates(*sharded._local_topk(qd, 100), 100)
    Q = s_loc.shape[0]
    os_, oi = torch.empty((Q, 100), device="cuda"), torch.empty((Q, 100), dtype=torch.int32, device="cuda")
    _lib.check(lib.fp_allgather_topk(ctx, _lib.ptr(s_loc.contiguous()), _lib.ptr(i_loc.contiguous()), Q, 100, 100, _lib.ptr(os_),
                                     _lib.ptr(oi), _lib.current_stream()), "fp_allgather_topk")
    torch.cuda.synchronize()
    assert torch.equal(oi.cpu(), i_full.cpu()) and torch.equal(os_.cpu(), s_full.cpu())
    _lib.check(lib.fp_comm_destroy(ctx), "fp_comm_destroy")
    if rank == 0:
        print("MULTIRANK_CABI_OK", world, flush=True)


def main():
    import os
    backend = os.environ.get("FP_DIST_BACKEND", "gloo")      # "nccl" (RCCL) when every rank has its own GPU
    rank, world, local = parallel.init_from_env(backend)
    assert world >= 2
    bank_f32 = bench.synthetic_bank(46037, 1024, seed=21)
    q = np.random.default_rng(9).standard_normal((6, 1024)).astype(np.float32)
    q[2] = bank_f32[40000] * 2.0                       # planted row lives in the last rank's shard
    qd = ops.l2_normalize(torch.from_numpy(q).cuda().to(torch.bfloat16))
    sharded = TemplateBank(bank_f32, shard=True)
    assert sharded.sharded and (sharded.lo, sharded.hi) == parallel.shard_range(46037, rank, world)
    s_sh, i_sh = sharded.topk(qd, 100)
    full = TemplateBank(bank_f32, shard=False)
    s_full, i_full = full.topk(qd, 100)
    assert torch.equal(i_sh.cpu(), i_full.cpu()) and torch.equal(s_sh.cpu(), s_full.cpu()), f"rank {rank}: sharded != unsharded"
    assert int(i_sh[2, 0]) == 40000
    # every rank must hold the same merged result
    gathered = parallel.all_gather_cat(i_sh.to(torch.int64).reshape(1, -1), dim=0)
    assert all(torch.equal(gathered[0], gathered[r]) for r in range(world))
    # variable-length float64 CUDA rows (the drivers' pose rows) and the frame-sharded soft vote
    rows = torch.tensor([[float(rank), 1.5 * j] for j in range(2 + rank)], dtype=torch.float64, device="cuda")
    allr = parallel.all_gather_rows(rows)
    assert allr.shape == (sum(2 + r for r in range(world)), 2) and allr.dtype == torch.float64
    assert sorted(allr[:, 0].tolist()) == sorted(float(r) for r in range(world) for _ in range(2 + r))
    n_frames, n_obj = 5, 2
    fq = np.random.default_rng(12).standard_normal((n_frames, n_obj, 1024)).astype(np.float32)
    fq[:, 0] = bank_f32[777] + 0.002 * fq[:, 0]          # object 0 looks like bank row 777 in every frame
    frame_q = [ops.l2_normalize(torch.from_numpy(fq[f]).cuda().to(torch.bfloat16)) for f in range(n_frames)]
    mine = parallel.shard_items(n_frames, rank, world)
    rows_sh, best_sh = full.soft_vote([frame_q[f] for f in mine], k=50, frame_ids=mine, n_obj=n_obj)   # (a rank without frames joins with 0 rows)
    votes = [full.frame_votes(q, 50) for q in frame_q]
    ref_rows, ref_best, _ = _soft_vote_local(votes, full.N)
    assert np.array_equal(rows_sh, ref_rows) and np.array_equal(best_sh, ref_best), (rows_sh, ref_rows, best_sh, ref_best)
    assert rows_sh[0] == 777, rows_sh
    if rank == 0:
        print("MULTIRANK_BANK_OK", world, flush=True)
    if backend == "nccl" and os.environ.get("FP_COMM_DIR"):
        _c_abi_comm(rank, world, Path(os.environ["FP_COMM_DIR"]), s_sh.shape, sharded, qd, s_full, i_full)
        # ONE comm stack (round 5): the same parallel.* calls with the library's communicator as the transport (FP_COMM_STACK=capi:
        # unique id carried by the process group, fp_allgather_bytes underneath) give what the torch.distributed transport gave
        parallel.use_capi_comm()
        assert parallel.comm_stack() == "capi"
        s2, i2 = sharded.topk(qd, 100)
        assert torch.equal(i2.cpu(), i_full.cpu()) and torch.equal(s2.cpu(), s_full.cpu())
        assert torch.equal(parallel.all_gather_rows(rows), allr)
        r2, b2 = full.soft_vote([frame_q[f] for f in mine], k=50, frame_ids=mine, n_obj=n_obj)
        assert np.array_equal(r2, ref_rows) and np.array_equal(b2, ref_best)
        if rank == 0:
            print("MULTIRANK_ONE_STACK_OK", world, flush=True)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
