"""Drop-in for the reference's `src/pipeline/estimators/online_pose_estimator.py` (DinoOnlinePoseEstimator :16-96).

Per (frame, object): coarse estimate on the first frame, then render-and-compare over the fine-grid rotations within
`neighborhood` degrees of the previous pose.  Everything between the proposal crop and the 4x4 pose stays on the GPU:
fp_geodesic_select -> fp_rasterize (all neighbours in one batch) -> fp_depth_extents + fp_crop_resize_pad ->
fp_vit_forward -> fp_template_score -> arg-max; only the winning index, its score and its two cloud extents return to
the host.  One ViT instance is shared with the coarse estimator (the reference loads two copies, :19-20).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from freepose_amd import ops
from freepose_amd.src.pipeline.estimators.pose_estimator import DinoPoseEstimator, _intrinsics
from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer
from freepose_amd.src.pipeline.utils import z_from_extents


class _HypothesisStore:
    """What the render-and-compare step computed for fine-grid hypotheses of ONE mesh (at one ViT layer): patch features, cloud extents
    and — when the scores are mask-weighted — the render masks, device-resident, addressed by fine-grid index.  A grid pose is the same
    render, the same crop and (a crop's features do not depend on its batch: tests/test_gpu_fuzz.py) the same feature bits in every
    frame, and consecutive frames share most of their 15-degree neighbourhood, so only the hypotheses that ENTER the neighbourhood are
    rendered and sent through the ViT.  The reference recomputes all of them per frame (online_pose_estimator.py:55-79); the results
    are identical.  When `cap` would be exceeded the store is emptied (the next step refills it with its own neighbourhood)."""

    def __init__(self, mesh, cap, need_masks):
        self.mesh, self.cap, self.need_masks = mesh, int(cap), bool(need_masks)    # (the mesh reference keeps id(mesh) unique)
        self.slot = {}
        self.feats = self.ext = self.masks = None

    def make_room(self, grid_ids) -> bool:
        """called once per step with ALL grid ids the step needs from this store: empties the store when they would not fit beside what
        it holds; False = they do not fit at all (the step then recomputes them without the store)"""
        need = {int(g) for g in grid_ids}
        if len(need) > self.cap:
            return False
        if len(need | set(self.slot)) > self.cap:
            self.slot.clear()
        return True

    def reserve(self, grid_ids):
        """slots for grid ids not held yet -> (ids, slots)"""
        new = [int(g) for g in grid_ids if int(g) not in self.slot]
        base = len(self.slot)
        for k, g in enumerate(new):
            self.slot[g] = base + k
        return new, list(range(base, base + len(new)))

    def _grow(self, rows, feats, ext, masks):
        """device buffers for at least `rows` hypotheses: 128 to start with, doubled (contents kept) up to `cap` — a clip that stays near
        its first pose never pays for the full store (1.8 MB per ViT-L hypothesis)"""
        have = 0 if self.feats is None else self.feats.shape[0]
        if rows <= have:
            return
        size = min(self.cap, max(128, 2 * have, rows))

        def grown(old, like):
            new = torch.empty((size,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
            if old is not None:
                new[:have] = old
            return new
        self.feats, self.ext = grown(self.feats, feats), grown(self.ext, ext)
        if self.need_masks:
            self.masks = grown(self.masks, masks)

    def write(self, slots, feats, ext, masks):
        self._grow(max(slots) + 1, feats, ext, masks)
        idx = torch.as_tensor(slots, dtype=torch.long, device=feats.device)
        self.feats.index_copy_(0, idx, feats)
        self.ext.index_copy_(0, idx, ext)
        if self.need_masks:
            self.masks.index_copy_(0, idx, masks)

    def gather(self, grid_ids):
        idx = torch.as_tensor([self.slot[int(g)] for g in grid_ids], dtype=torch.long, device=self.feats.device)
        return self.feats.index_select(0, idx), self.ext.index_select(0, idx), (self.masks.index_select(0, idx) if self.need_masks else None)


class DinoOnlinePoseEstimator(torch.nn.Module):
    def __init__(self, n_coarse_poses=600, n_fine_poses=10000, cache_size=50, save_all=False, cache_dir="./data/cache",
                 feature_extractor=None, hypothesis_cache=768, hypothesis_meshes=8):
        """`hypothesis_cache`: fine-grid hypotheses kept per mesh between frames (_HypothesisStore; 0 = recompute every hypothesis in
        every frame like the reference — same results); `hypothesis_meshes`: meshes that keep such a store (least recently used out).
        768 hypotheses of a ViT-L @420^2 are 1.4 GB (the buffers start at 128 hypotheses and double as the object turns)."""
        super().__init__()
        self.hypothesis_cache, self.hypothesis_meshes = int(hypothesis_cache), int(hypothesis_meshes)
        self._hyp_stores = OrderedDict()             # (id(mesh), layer, masks?) -> _HypothesisStore
        self._nb_cache, self._grid_keys = {}, None   # neighbourhood of a fine-grid rotation by its bytes (_neighbourhood)
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

    def _neighbourhood(self, R_prev, thresh_deg):
        """fp_geodesic_select of the previous pose.  From the second tracked frame on the previous pose IS a fine-grid rotation (the winner
        of the last step, copied into TCO), so its neighbourhood is a function of that grid index: looked up by the rotation's bytes, computed
        once per (grid rotation, threshold) — a kernel, a compaction and two device -> host copies less per frame and object."""
        R = np.ascontiguousarray(R_prev, dtype=np.float64)
        key = (R.tobytes(), thresh_deg)
        hit = self._nb_cache.get(key)
        if hit is not None:
            return hit
        close = ops.geodesic_select(self._fine_rots_dev, R, thresh_deg)
        if self._grid_keys is None:
            self._grid_keys = {np.ascontiguousarray(P[:3, :3], dtype=np.float64).tobytes() for P in self.fine_mesh_poses}
        if key[0] in self._grid_keys:                 # (arbitrary rotations — the coarse estimate of frame 0 — are not worth remembering)
            if len(self._nb_cache) >= 65536:
                self._nb_cache.clear()
            self._nb_cache[key] = close
        return close

    def _hypothesis_store(self, mesh, layer, need_masks):
        """the store of a mesh (created on first use, most recently used last).  Nothing is evicted here: a step resolves the stores of
        ALL its objects first and trims afterwards (_trim_stores), so a frame that tracks more distinct meshes than `hypothesis_meshes`
        keeps every store it is using for the duration of the step."""
        key = (id(mesh), int(layer), bool(need_masks), float(self.rendering_scale))
        st = self._hyp_stores.get(key)
        if st is None:
            st = self._hyp_stores[key] = _HypothesisStore(mesh, self.hypothesis_cache, need_masks)
        self._hyp_stores.move_to_end(key)
        return st

    def _trim_stores(self, in_use=0):
        """least recently used stores out, down to max(hypothesis_meshes, stores the current step uses)"""
        bound = max(1, self.hypothesis_meshes, int(in_use))
        while len(self._hyp_stores) > bound:
            self._hyp_stores.popitem(last=False)

    def forward_fine(self, proposal, proposal_mask, template_dict, mesh, K, bbox, est_scale, prev_pose, neighborhood=15,
                     layer=22, mask_scores=False, query_feat=None):
        item = dict(proposal=proposal, proposal_mask=proposal_mask, template_dict=template_dict, mesh=mesh, K=K, bbox=bbox,
                    est_scale=est_scale, prev_pose=prev_pose, query_feat=query_feat)
        return self.forward_fine_many([item], neighborhood, layer, mask_scores)[0]

    def forward_fine_many(self, items, neighborhood=15, layer=22, mask_scores=False):
        """The render-and-compare step (reference :43-96) for SEVERAL (frame, object) pairs at once — the objects of one video
        frame, which the reference visits one after the other (scripts/dino_inference_video.py:124-156) and which do not depend on
        each other.  Each object is rendered from its own mesh at its own neighbour poses; all hypothesis crops and all query crops
        then go through ONE ViT call (a crop's features do not depend on its batch neighbours — bit-exact, tested — and ~20-crop
        batches leave the 256-CU GEMM grids a third empty), every object is scored against its own query, and the winners of all
        objects return in ONE device -> host copy.  `items`: dicts with proposal, proposal_mask, template_dict, mesh, K, bbox,
        est_scale, prev_pose and optionally query_feat (frame 0: the coarse estimator's un-normalised query, reference :40-41).
        Returns one reference-style result dict per item, identical to calling `forward_fine` per item."""
        if len(items) == 0:
            return []
        work, pieces = [], []
        neighbourhoods, stores, wanted = [], [], {}
        for it in items:
            close = self._neighbourhood(np.asarray(it["prev_pose"])[:3, :3], float(neighborhood))
            if len(close) == 0:
                raise RuntimeError("no fine-grid rotation within the neighbourhood of the previous pose")
            neighbourhoods.append(close)
            st = self._hypothesis_store(it["mesh"], layer, mask_scores) if self.hypothesis_cache > 0 else None
            stores.append(st)                          # resolved ONCE per item: the second loop must see the same object
            if st is not None:
                wanted.setdefault(id(st), (st, []))[1].extend(int(g) for g in close)
        self._trim_stores(in_use=len(wanted))          # (after the step's stores were all touched: none of them is the LRU victim)
        usable = {k: st.make_room(ids) for k, (st, ids) in wanted.items()}       # (objects that share a mesh share its store)
        for it, close, store in zip(items, neighbourhoods, stores):
            if store is not None and not usable[id(store)]:
                store = None
            todo, slots = store.reserve(close) if store is not None else ([int(g) for g in close], None)
            crops = masks = ext = None
            if todo:                                   # hypotheses not seen for this mesh yet (all of them without a store)
                renders = self.renderer.render_from_poses(it["mesh"], self.fine_mesh_poses[todo], scale=self.rendering_scale,
                                                          depth=mask_scores)   # (masks are depth > 0; extents come from the tile epilogue)
                crops, _, masks, ext = MeshRenderer.generate_proposals(renders, out_bf16=True, return_extents=True, need_masks=mask_scores)
            proposal, query_feat = it["proposal"], it.get("query_feat")
            # the query crop rides in the same ViT batch as the hypothesis crops: a separate B = 1 forward is launch-bound
            # (~2.5 ms of a ~19 ms step)
            rides = query_feat is None and tuple(proposal.shape[-2:]) == (420, 420)
            if rides:
                pieces.append(torch.as_tensor(proposal)[None].to("cuda", torch.bfloat16))
            elif query_feat is None:
                query_feat = ops.l2_normalize(self.feature_extractor(proposal[None], layer=layer, feature_type="patch"))
            if todo:
                pieces.append(crops)
            work.append(dict(close=close, n_new=len(todo), slots=slots, store=store, masks=masks, ext=ext, rides=rides, query_feat=query_feat))
        feats_all = self.feature_extractor(pieces[0] if len(pieces) == 1 else torch.cat(pieces, dim=0), layer=layer,
                                           feature_type="patch") if pieces else None
        at = 0
        for w in work:                                 # first every store receives its new hypotheses (two objects may share a mesh) ...
            if w["rides"]:
                w["query_feat"] = ops.l2_normalize(feats_all[at:at + 1])
                at += 1
            w["feats"] = feats_all[at:at + w["n_new"]] if w["n_new"] else None
            at += w["n_new"]
            if w["store"] is not None and w["n_new"]:
                w["store"].write(w["slots"], w["feats"], w["ext"], w["masks"])
        packed = []
        for it, w in zip(items, work):                 # ... then every object is scored against its own neighbourhood, in grid order
            n = len(w["close"])
            if w["store"] is not None:
                feats, ext, masks = w["store"].gather(w["close"])
            else:
                feats, ext, masks = w["feats"], w["ext"], w["masks"]
            w["ext"] = ext
            q = w["query_feat"].reshape(-1, w["query_feat"].shape[-1])
            weights = None
            if mask_scores:
                m = torch.logical_or(masks, torch.as_tensor(it["proposal_mask"]).to(masks.device)[None]).float()
                g = int(round(feats.shape[1] ** 0.5))
                weights = torch.nn.functional.interpolate(m[None], size=(g, g), mode="bilinear")[0].reshape(n, -1)
            scores = ops.template_score(feats, q, weights)
            # max / argmax (first maximum): canonical (score desc, index asc)
            idx_all = torch.arange(n, dtype=torch.int32, device=scores.device)
            top_s, top_i = ops.topk_merge(scores[None], idx_all[None], 1)
            # winner index, its score and its two cloud extents (the score is a bf16 value held in fp32, the index is small: both
            # are exact in float64)
            packed.append(torch.cat([top_i[0, :1].double(), top_s[0, :1].double(), ext[top_i[0, 0].long(), 4:6].double()]))
        packed = torch.stack(packed).cpu().numpy()          # ONE device -> host copy for all objects of the frame
        outs = []
        for it, w, pk in zip(items, work, packed):
            top = int(pk[0])
            ratio = float(it["est_scale"]) / 0.25
            TCO = z_from_extents(it["bbox"], pk[2] * ratio, pk[3] * ratio, it["K"], self.fine_mesh_poses[int(w["close"][top])])
            outs.append({"TCO": [TCO], "scores": [np.float32(pk[1])], "proposal": it["proposal"], "K": it["K"], "bbox": it["bbox"]})
        return outs
