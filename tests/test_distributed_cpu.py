"""world_size-2 gloo tests of the N>1 path (freepose_amd/parallel.py): bank-row sharding + all-gather + canonical merge
gives the unsharded top-k on every rank; proposal/object sharding + row all-gather reassembles every result.  The
per-shard scan is played by the oracle here (no GPU); on the MI355X the same code runs fp_bank_topk / fp_topk_merge over
RCCL."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from freepose_amd import parallel
    from oracle import fp_oracle as fo
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world) and parallel.world() == (rank, world)

    # ---- bank-row sharding -------------------------------------------------------------
    N, D, Q, k = 1501, 384, 5, 100
    rng = np.random.Generator(np.random.PCG64(11))
    base = rng.standard_normal((40, D)).astype(np.float32)
    bank = base[rng.integers(0, 40, N)] + 0.01 * rng.standard_normal((N, D)).astype(np.float32)  # tie-heavy
    bank_bits = fo.bank_prepare(bank)
    q_bits = fo.l2norm_rows(fo.to_bf16_bits(rng.standard_normal((Q, D)).astype(np.float32)))
    lo, hi = parallel.shard_range(N, rank, world)
    assert (lo, hi) == ((0, 751), (751, 1501))[rank]

    def local(queries, kk):
        s, i = fo.bank_topk(bank_bits[lo:hi], q_bits, min(kk, hi - lo), idx_offset=lo)
        return torch.from_numpy(s), torch.from_numpy(i)
    s, i = parallel.sharded_bank_topk(local, None, k)
    s_ref, i_ref = fo.bank_topk(bank_bits, q_bits, k)
    assert np.array_equal(i.numpy(), i_ref) and np.array_equal(s.numpy(), s_ref)

    # k larger than the smaller shard (751 vs 750 rows): per-rank lists are padded with sentinels to equal shapes
    s2, i2 = parallel.sharded_bank_topk(local, None, 751)
    s2_ref, i2_ref = fo.bank_topk(bank_bits, q_bits, 751)
    assert np.array_equal(i2.numpy(), i2_ref) and np.array_equal(s2.numpy(), s2_ref)
    assert int(i2.max()) < N and bool(torch.isfinite(s2).all())

    # ---- video soft-vote as a sharded reduction (extract_proposals_ground_video.py:154-159,186-190) -------------
    n_frames, n_obj, kk = 7, 3, 20
    fr_q = fo.l2norm_rows(fo.to_bf16_bits(rng.standard_normal((n_frames * n_obj, D)).astype(np.float32))).reshape(n_frames, n_obj, D)
    lists = [fo.bank_topk(bank_bits, fr_q[f], kk) for f in range(n_frames)]
    mine = parallel.shard_items(n_frames, rank, world)
    best_row, best, mean = parallel.soft_vote_reduce(torch.from_numpy(np.stack([lists[f][0] for f in mine])),
                                                     torch.from_numpy(np.stack([lists[f][1] for f in mine])),
                                                     torch.tensor(mine, dtype=torch.int64), N)
    dense = np.zeros((n_frames, n_obj, N), np.float32)      # the reference's dense formulation, frame order, float32
    for f in range(n_frames):
        for o in range(n_obj):
            dense[f, o, lists[f][1][o]] = lists[f][0][o]
    acc = np.zeros((n_obj, N), np.float32)
    for f in range(n_frames):
        acc = acc + dense[f]
    acc = acc / np.float32(n_frames)
    assert np.array_equal(mean.numpy(), acc)
    assert np.array_equal(best_row.numpy(), acc.argmax(axis=1)) and np.array_equal(best.numpy(), acc.max(axis=1))

    # a clip with fewer frames than ranks: the rank without a frame joins the collective with zero rows (it used to raise in
    # torch.stack([]) while the other ranks waited in the all-gather)
    one = [0] if rank == 0 else []
    br1, b1, m1 = parallel.soft_vote_reduce(torch.from_numpy(np.stack([lists[f][0] for f in one]).reshape(len(one), n_obj, kk)),
                                            torch.from_numpy(np.stack([lists[f][1] for f in one]).reshape(len(one), n_obj, kk))
                                            if one else torch.zeros((0, n_obj, kk), dtype=torch.int32),
                                            torch.tensor(one, dtype=torch.int64), N) if one else \
        parallel.soft_vote_reduce(torch.zeros((0, n_obj, kk)), torch.zeros((0, n_obj, kk), dtype=torch.int32),
                                  torch.zeros((0,), dtype=torch.int64), N)
    assert np.array_equal(m1.numpy(), dense[0]) and np.array_equal(br1.numpy(), dense[0].argmax(axis=1))

    # ---- proposal / object sharding: round-robin items, variable rows per rank ---------------------------
    items = parallel.shard_items(7, rank, world)
    assert items == ([0, 2, 4, 6], [1, 3, 5])[rank]
    rows = torch.tensor([[float(it), it * 10.0, rank] for it in items], dtype=torch.float64)
    allr = parallel.all_gather_rows(rows)
    assert allr.shape == (7, 3) and sorted(allr[:, 0].tolist()) == list(range(7))
    assert torch.equal(parallel.all_gather_cat(torch.full((2, 3), float(rank)), dim=1)[:, ::3], torch.tensor([[0., 1.], [0., 1.]]))
    # ---- the "who ran where" record every N-rank line carries (bench.py, the CLIs) ---------------------------
    rep = parallel.rank_report()
    assert rep["backend"] == "gloo" and rep["world_size"] == 2 and [r["rank"] for r in rep["ranks"]] == [0, 1]
    assert rep["shared_devices"] is False and all(r["device"] is None for r in rep["ranks"])   # CPU ranks: no device to share
    dist.barrier()
    Path(tmp, f"ok{rank}").write_text("ok")
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_single_process_paths():
    from freepose_amd import parallel
    assert parallel.world() == (0, 1)
    assert parallel.shard_range(10, 0, 1) == (0, 10)
    assert [parallel.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    t = torch.arange(6.0).reshape(2, 3)
    assert parallel.all_gather_cat(t) is t and parallel.all_gather_rows(t) is t
    s = torch.tensor([[0.5, 0.75, 0.75, 0.25]])
    i = torch.tensor([[7, 9, 3, 1]], dtype=torch.int32)
    ms, mi = parallel.merge_topk(s, i, 3)
    assert mi.tolist() == [[3, 9, 7]] and ms.tolist() == [[0.75, 0.75, 0.5]]
