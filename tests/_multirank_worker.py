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


def main():
    rank, world, local = parallel.init_from_env("gloo")
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
    if rank == 0:
        print("MULTIRANK_BANK_OK", world, flush=True)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
