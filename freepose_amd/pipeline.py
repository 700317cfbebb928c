"""The per-proposal hot path as one batched, device-resident pipeline (what bench.py times and smoke() checks):

    crop [3,R,R] + mask  ->  ViT-L/14 layer-22 patch features  ->  FFA descriptor  ->  cosine top-k over the bank
                         ->  H pose hypotheses: rasterise -> mask bbox -> crop/resize -> ViT -> patchwise score vs the query
                         ->  top-3 hypotheses -> metric (R, t) from the render's depth extents

It is the composition the reference performs per proposal across scripts/extract_proposals_ground.py:126-145 (retrieval)
and src/pipeline/estimators/{pose_estimator.py:79-118, online_pose_estimator.py:49-96} (render-and-compare), with the
template features computed from fresh renders (the reference's cache-miss / online path) so no stage is skipped.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from freepose_amd import ops, parallel
from freepose_amd.retrieval import TemplateBank
from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer, grid_poses
from freepose_amd.src.pipeline.utils import z_from_extents


class StageClock:
    """HIP-event stopwatch per named stage (bench.py's per-stage table): intervals are recorded on the current stream with
    no host synchronisation; read() resolves them after the timed region."""

    def __init__(self):
        self._spans = []      # (name, Timer)
        self._pool = []

    class _Span:
        def __init__(self, clock, name):
            self.clock, self.name = clock, name

        def __enter__(self):
            self.t = self.clock._pool.pop() if self.clock._pool else ops.Timer()
            self.t.start()

        def __exit__(self, *exc):
            self.t.stop()
            self.clock._spans.append((self.name, self.t))

    def stage(self, name: str):
        return StageClock._Span(self, name)

    def read(self) -> dict:
        """{stage: total ms} since the last read"""
        out = {}
        for name, t in self._spans:
            out[name] = out.get(name, 0.0) + t.elapsed_ms()
            self._pool.append(t)
        self._spans = []
        return out


class _NoClock:
    class _Null:
        def __enter__(self):
            return None

        def __exit__(self, *exc):
            return False

    _null = _Null()

    def stage(self, name):
        return self._null


@dataclass
class ProposalResult:
    topk_scores: np.ndarray      # [k] retrieval scores
    topk_idx: np.ndarray         # [k] bank rows
    hyp_idx: np.ndarray          # [3] best hypotheses
    hyp_scores: np.ndarray       # [3]
    TCO: List[np.ndarray]        # 3 x [4,4]


class HotPath:
    def __init__(self, vit: ops.ViT, bank: TemplateBank, mesh: ops.Mesh, n_hyp: int = 576, crop_res: int = 518,
                 render_res: int = 420, k: int = 100, layer: int = 22, vit_batch: int = 192, render_scale: float = 0.25):
        self.vit, self.bank, self.mesh = vit, bank, mesh
        self.n_hyp, self.crop_res, self.render_res, self.k, self.layer = n_hyp, crop_res, render_res, k, layer
        self.vit_batch, self.render_scale = vit_batch, render_scale
        self.hyp_poses = np.array(grid_poses(n_hyp))
        self._poses_dev = torch.from_numpy(self.hyp_poses.astype(np.float32)).cuda()
        self.fx = self.fy = 600.0 * render_res / 420.0
        self.cx = self.cy = render_res / 2
        self.clock = _NoClock()      # bench.py installs a StageClock

    def retrieve(self, crops: torch.Tensor, masks: torch.Tensor):
        """crops bf16 [B,3,R,R], masks bool/u8 [B,R,R] -> (query patch feats [B,P,D], top-k scores, top-k idx)"""
        with self.clock.stage("vit_query"):
            feats = self.vit(crops, layer=self.layer, feature_type="patch")
        g = self.crop_res // 14
        with self.clock.stage("ffa"):
            desc = ops.ffa(feats, masks[:, : g * 14, : g * 14], cell=14, normalize=True)
        with self.clock.stage("bank_scan_topk"):
            s, i = self.bank.topk(desc, self.k)
        return feats, s, i

    def render_hypotheses(self):
        # the depth image is never written: its two consumers (depth>0 box, cloud extents) are reduced in the rasteriser's tile epilogue
        with self.clock.stage("rasterize"):
            rgb, _, ext, boxes = ops.rasterize_extents(self.mesh, self._poses_dev, self.render_scale, self.fx, self.fy, self.cx, self.cy,
                                                       self.render_res, self.render_res)
        with self.clock.stage("crop_resize"):
            crops = ops.crop_resize_pad(rgb, boxes, self.crop_res, 0.0, out_bf16=True)
        return crops, ext

    def hypothesis_features(self, crops: torch.Tensor, normalized: bool = False) -> torch.Tensor:
        """patch features of the hypothesis crops; `normalized`: rows F.normalize()d by the ViT's final-norm kernel (what run() scores
        with: the streaming template_dots_normed kernel then reads them once, same bits as normalising on the fly)"""
        g = self.crop_res // 14
        sizes = ops.plan_vit_batches(crops.shape[0], g * g + 5, self.vit_batch)   # whole GEMM tile rounds per batch
        with self.clock.stage("vit_hypotheses"):
            feats = torch.empty((crops.shape[0], g * g, self.vit.dim), dtype=torch.bfloat16, device=crops.device)
            i = 0
            for b in sizes:      # each chunk writes its slice of the result: no concatenation pass (1.6 GB for 576 crops)
                self.vit(crops[i:i + b], layer=self.layer, feature_type="patch_normalized" if normalized else "patch", out=feats[i:i + b])
                i += b
            return feats

    def run(self, crops: torch.Tensor, masks: torch.Tensor, K: np.ndarray, bboxes: np.ndarray, scales) -> List[ProposalResult]:
        """one pass of the hot path over a batch of proposals (every stage executed for every proposal)"""
        feats, top_s, top_i = self.retrieve(crops, masks)
        out = []
        for b in range(crops.shape[0]):
            hyp_crops, ext = self.render_hypotheses()          # the retrieved mesh under all hypotheses
            hyp_feats = self.hypothesis_features(hyp_crops, normalized=True)
            with self.clock.stage("template_score"):
                q = ops.l2_normalize(feats[b])
                scores = ops.template_score(hyp_feats, q, normalized=True)
            with self.clock.stage("hypothesis_top3"):
                idx_all = torch.arange(self.n_hyp, dtype=torch.int32, device=scores.device)
                s3, i3 = ops.topk_merge(scores[None], idx_all[None], 3)
            # everything the host needs of this proposal in ONE device -> host copy (five separate .cpu() calls were five
            # synchronisations with the GPU idle in between): indices and fp32 scores are exact in float64
            k = top_s.shape[1]
            packed = torch.cat([i3[0].double(), s3[0].double(), ext[i3[0].long(), 4:6].double().reshape(-1), top_s[b].double(),
                                top_i[b].double()]).cpu().numpy()
            i3h = packed[0:3].astype(np.int64)
            s3h = packed[3:6].astype(np.float32)
            e = packed[6:12].reshape(3, 2)
            ratio = float(scales[b]) / self.render_scale
            tco = [z_from_extents(bboxes[b], e[j, 0] * ratio, e[j, 1] * ratio, K, self.hyp_poses[i3h[j]]) for j in range(3)]
            out.append(ProposalResult(packed[12:12 + k].astype(np.float32), packed[12 + k:12 + 2 * k].astype(np.int32), i3h, s3h, tco))
        return out


def pack_results(results: List[ProposalResult]) -> torch.Tensor:
    """fixed-width rows (top-1 mesh row, score, best hypothesis, score, R9, t3) for the result all-gather"""
    rows = [[float(r.topk_idx[0]), float(r.topk_scores[0]), float(r.hyp_idx[0]), float(r.hyp_scores[0]),
             *r.TCO[0][:3, :3].flatten().tolist(), *r.TCO[0][:3, 3].tolist()] for r in results]
    return torch.tensor(rows, dtype=torch.float64).reshape(-1, 16)
