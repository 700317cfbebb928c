"""Mesh retrieval against the template bank (reference: scripts/extract_proposals_ground.py:39-44,136-160 and
scripts/extract_proposals_ground_video.py:149-190).  The bank stays resident in HBM as normalised bf16 rows; a query
batch is one coalesced scan (fp_bank_topk).  With world_size > 1 the bank rows are sharded across ranks and the per-rank
top-k lists are all-gathered over RCCL and merged (freepose_amd.parallel)."""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch

from freepose_amd import ops, parallel


class TemplateBank:
    def __init__(self, bank_f32: np.ndarray | torch.Tensor, mesh_ids: Optional[Sequence[str]] = None, shard: bool = False):
        """bank_f32 [N,D] fp32 (data/<folder>.npy); row i <-> mesh_ids[i] (data/mesh_cache.txt)."""
        bank = torch.as_tensor(bank_f32, dtype=torch.float32)
        self.N, self.D = bank.shape
        self.mesh_ids = list(mesh_ids) if mesh_ids is not None else None
        if self.mesh_ids is not None and len(self.mesh_ids) != self.N:
            raise ValueError(f"bank has {self.N} rows but the file list has {len(self.mesh_ids)} ids (SURVEY App. A-5)")
        rank, ws = parallel.world()
        self.lo, self.hi = parallel.shard_range(self.N, rank, ws) if (shard and ws > 1) else (0, self.N)
        self.sharded = shard and ws > 1
        self.rows = ops.bank_prepare(bank[self.lo:self.hi])   # cast -> bf16 -> bf16 row normalise (ground.py:40-41)

    @classmethod
    def from_files(cls, npy_path, filelist_path=None, shard=False):
        ids = Path(filelist_path).read_text().splitlines() if filelist_path else None
        return cls(np.load(npy_path), ids, shard)

    def _local_topk(self, queries, k):
        kk = min(k, self.hi - self.lo)          # a shard may hold fewer than k rows; sharded_bank_topk pads to k
        return ops.bank_topk(self.rows, queries, kk, idx_offset=self.lo)

    def topk(self, queries: torch.Tensor, k: int = 100):
        """queries bf16 [Q,D] (already F.normalize'd FFA descriptors) -> (scores f32 [Q,k], idx i32 [Q,k]); k is clamped to
        the bank size (torch.topk would raise; the reference always asks for 100 of 46 037)"""
        k = min(int(k), self.N)
        if self.sharded:
            return parallel.sharded_bank_topk(self._local_topk, queries, k)
        return self._local_topk(queries, k)

    def retrieve(self, queries: torch.Tensor):
        """topk == 0 path of the reference: (mesh id, score) of the best row per query (ground.py:142-145)."""
        s, i = self.topk(queries, min(100, self.N))
        s, i = s[:, 0].cpu().numpy(), i[:, 0].cpu().numpy()
        names = [self.mesh_ids[j] if self.mesh_ids else int(j) for j in i]
        return names, s.tolist(), i

    # ---- per-view fine re-rank (extract_proposals_ground.py:147-160; --topk k) -----------------------------------------
    def attach_views(self, per_mesh_views: Sequence[np.ndarray]):
        """per_mesh_views[i]: the [<=600, D] fp32 descriptor file of bank row i (data/datasets/<retrieval>/<mesh>.npy).
        Kept device-resident as raw bf16 rows back to back (the whole Objaverse-LVIS+GSO store is 46037 x 600 x 1024 x 2 B
        = 56.6 GB, inside one MI355X's HBM) instead of 100 np.load calls per proposal."""
        if len(per_mesh_views) != self.N:
            raise ValueError("need one per-view descriptor array per bank row")
        counts = np.array([len(v) for v in per_mesh_views], dtype=np.int64)
        self.view_offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).cuda()
        self.views = torch.cat([torch.as_tensor(v, dtype=torch.float32) for v in per_mesh_views]).to("cuda", torch.bfloat16)

    def retrieve_reranked(self, queries: torch.Tensor, topk: int = 25, n_coarse: int = 100):
        """coarse top-100 scan, then per-candidate mean of its top-k per-view scores; the winner is the FIRST maximum in
        coarse order (Python `max(dict, key=dict.get)` semantics).  Returns (names, scores, bank rows, fine scores [Q,C])."""
        s, i = self.topk(queries, min(n_coarse, self.N))
        fine = ops.rerank_views(self.views, self.view_offsets, i, queries, topk)
        pos = torch.arange(fine.shape[1], dtype=torch.int32, device=fine.device)[None].expand(fine.shape[0], -1).contiguous()
        best_s, best_p = ops.topk_merge(fine, pos, 1)
        rows = torch.gather(i, 1, best_p.long())[:, 0].cpu().numpy()
        names = [self.mesh_ids[j] if self.mesh_ids else int(j) for j in rows]
        return names, best_s[:, 0].cpu().numpy().tolist(), rows, fine

    def frame_votes(self, queries: torch.Tensor, k: int = 100, topk: int = 0):
        """sparse soft-vote contribution of ONE frame: (scores f32 [n_obj,k], rows i32 [n_obj,k]) — the coarse top-k scores
        (topk == 0, ground_video.py:155-159) or each candidate's mean top-`topk` per-view score (:160-170)"""
        s, i = self.topk(queries, k)
        if topk:
            s = ops.rerank_views(self.views, self.view_offsets, i, queries, topk)
        return s, i

    def soft_vote(self, per_frame_queries: List[torch.Tensor], k: int = 100, topk: int = 0, frame_ids: Optional[Sequence[int]] = None,
                  n_obj: int = 0):
        """video soft-vote (ground_video.py:154-159,186-190): dense [N] score vectors with only each frame's top-k filled,
        mean over frames, per-object arg-max.  per_frame_queries[f] is bf16 [n_obj, D].  With `frame_ids` (the global numbers
        of the frames THIS rank holds) the vote is a collective: sparse lists are all-gathered and every rank reduces them in
        frame order (parallel.soft_vote_reduce) — identical result on all ranks and to a single-rank run.
        Returns (best row per object, its mean score)."""
        if self.sharded and frame_ids is not None:
            raise ValueError("soft_vote: a row-sharded bank makes topk() itself a collective over ALL ranks with the SAME queries; "
                             "it cannot be combined with frame sharding (different queries per rank) — replicate the bank")
        votes = [self.frame_votes(q, k, topk) for q in per_frame_queries]
        if votes:
            s = torch.stack([v[0] for v in votes])
            i = torch.stack([v[1] for v in votes])
        else:                   # a rank that holds no frame (fewer frames than ranks) still joins the collective, with 0 rows
            if not n_obj:
                raise ValueError("soft_vote: a rank without frames must pass n_obj (objects per frame)")
            s = torch.zeros((0, n_obj, k), dtype=torch.float32, device="cuda")
            i = torch.zeros((0, n_obj, k), dtype=torch.int32, device="cuda")
        fid = torch.arange(len(votes), dtype=torch.int64) if frame_ids is None else torch.as_tensor(list(frame_ids), dtype=torch.int64)
        if frame_ids is None and parallel.world()[1] > 1:
            raise ValueError("soft_vote under torch.distributed needs frame_ids (which frames this rank holds)")
        rows, best, _ = parallel.soft_vote_reduce(s, i, fid, self.N)
        return rows.cpu().numpy(), best.cpu().numpy()
