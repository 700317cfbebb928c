"""Drop-in for the reference's `src/pipeline/estimators/online_pose_estimator.py` (DinoOnlinePoseEstimator :16-96).

Per (frame, object): coarse estimate on the first frame, then render-and-compare over the fine-grid rotations within
`neighborhood` degrees of the previous pose.  Everything between the proposal crop and the 4x4 pose stays on the GPU:
fp_geodesic_select -> fp_rasterize (all neighbours in one batch) -> fp_depth_extents + fp_crop_resize_pad ->
fp_vit_forward -> fp_template_score -> arg-max; only the winning index, its score and its two cloud extents return to
the host.  One ViT instance is shared with the coarse estimator (the reference loads two copies, :19-20).
"""
from __future__ import annotations

import numpy as np
import torch

from freepose_amd import ops
from freepose_amd.src.pipeline.estimators.pose_estimator import DinoPoseEstimator, _intrinsics
from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer
from freepose_amd.src.pipeline.utils import z_from_extents


class DinoOnlinePoseEstimator(torch.nn.Module):
    def __init__(self, n_coarse_poses=600, n_fine_poses=10000, cache_size=50, save_all=False, cache_dir="./data/cache",
                 feature_extractor=None):
        super().__init__()
        self.coarse_estimator = DinoPoseEstimator(n_coarse_poses, cache_size, save_all, cache_dir, feature_extractor)
        self.feature_extractor = self.coarse_estimator.feature_extractor
        self.fine_mesh_poses = np.array(self.coarse_estimator.generate_poses(n_fine_poses))
        self._fine_rots_dev = torch.from_numpy(np.ascontiguousarray(self.fine_mesh_poses[:, :3, :3])).cuda()
        self.renderer = MeshRenderer(0)
        self.renderer.mesh_poses = list(self.fine_mesh_poses)
        self.rendering_scale = 0.25

    @staticmethod
    def geodesic_distance(render_poses, query_pose, degrees=True):
        """angle of R_i R_q^T (host numpy; API parity with the reference — the hot path uses fp_geodesic_select)."""
        R = np.asarray(render_poses)[:, :3, :3] @ np.asarray(query_pose)[:3, :3].T
        cosv = 0.5 * (np.trace(R, axis1=1, axis2=2) - 1.0)
        sinv = 0.5 * np.sqrt((R[:, 2, 1] - R[:, 1, 2]) ** 2 + (R[:, 0, 2] - R[:, 2, 0]) ** 2 + (R[:, 1, 0] - R[:, 0, 1]) ** 2)
        d = np.arctan2(sinv, cosv)
        return np.rad2deg(d) if degrees else d

    def forward(self, proposal, proposal_mask, template_dict, mesh, K, bbox, est_scale, prev_pose=None, neighborhood=15,
                layer=22, batch_size=128, mask_scores=False):
        if prev_pose is None:
            coarse = self.coarse_estimator.forward(proposal, template_dict, K, bbox, est_scale, layer, batch_size,
                                                   return_query_feat=True)
            query_feat = coarse["query_feat"]   # un-normalised on frame 0, as in the reference (:40-41; SURVEY A-3)
            prev_pose = coarse["TCO"][0]
        else:
            query_feat = None
        return self.forward_fine(proposal, proposal_mask, template_dict, mesh, K, bbox, est_scale, prev_pose, neighborhood,
                                 layer, mask_scores, query_feat)

    def forward_fine(self, proposal, proposal_mask, template_dict, mesh, K, bbox, est_scale, prev_pose, neighborhood=15,
                     layer=22, mask_scores=False, query_feat=None):
        close = ops.geodesic_select(self._fine_rots_dev, np.asarray(prev_pose)[:3, :3], float(neighborhood))
        if len(close) == 0:
            raise RuntimeError("no fine-grid rotation within the neighbourhood of the previous pose")
        selected = self.fine_mesh_poses[close]
        renders = self.renderer.render_from_poses(mesh, selected, scale=self.rendering_scale)
        crops, poses, masks, ext = MeshRenderer.generate_proposals(renders, out_bf16=True, return_extents=True, need_masks=mask_scores)
        if query_feat is None and tuple(proposal.shape[-2:]) == tuple(crops.shape[-2:]):
            # the query crop rides in the same ViT batch as the hypothesis crops: a separate B = 1 forward is launch-bound
            # (~2.5 ms of a ~19 ms step) and a crop's features do not depend on its batch neighbours (bit-exact, tested)
            both = torch.cat([torch.as_tensor(proposal)[None].to(crops.device, crops.dtype), crops], dim=0)
            both_feats = self.feature_extractor(both, layer=layer, feature_type="patch")
            query_feat, feats = ops.l2_normalize(both_feats[:1]), both_feats[1:]
        else:
            if query_feat is None:
                query_feat = ops.l2_normalize(self.feature_extractor(proposal[None], layer=layer, feature_type="patch"))
            feats = self.feature_extractor(crops, layer=layer, feature_type="patch")
        q = query_feat.reshape(-1, query_feat.shape[-1])
        weights = None
        if mask_scores:
            m = torch.logical_or(masks, torch.as_tensor(proposal_mask).to(masks.device)[None]).float()
            g = int(round(feats.shape[1] ** 0.5))
            weights = torch.nn.functional.interpolate(m[None], size=(g, g), mode="bilinear")[0].reshape(len(close), -1)
        scores = ops.template_score(feats, q, weights)
        # max / argmax (first maximum): canonical (score desc, index asc)
        idx_all = torch.arange(len(close), dtype=torch.int32, device=scores.device)
        top_s, top_i = ops.topk_merge(scores[None], idx_all[None], 1)
        # winner index, its score and its two cloud extents come back in ONE device -> host copy (the score is a bf16 value held
        # in fp32, the index is small: both are exact in float64)
        packed = torch.cat([top_i[0, :1].double(), top_s[0, :1].double(), ext[top_i[0, 0].long(), 4:6].double()]).cpu().numpy()
        top = int(packed[0])
        ratio = float(est_scale) / 0.25
        TCO = z_from_extents(bbox, packed[2] * ratio, packed[3] * ratio, K, poses[top])
        return {"TCO": [TCO], "scores": [np.float32(packed[1])], "proposal": proposal, "K": K, "bbox": bbox}
