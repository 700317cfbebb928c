"""How much of the video step the hypothesis store saves as a function of how fast the object turns: the bench's one-object clip
(bench.video_workload) at several rotation speeds, with the store and with every hypothesis recomputed per frame (the reference's step).
The fine grid is ~9 degrees apart and the neighbourhood 15 degrees wide, so a slow object re-uses almost everything and a fast one little.
    python tools/video_store_curve.py [frames]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from freepose_amd import ops  # noqa: E402
from freepose_amd.mesh_io import TriMesh  # noqa: E402
from freepose_amd.src.pipeline.estimators.online_pose_estimator import DinoOnlinePoseEstimator  # noqa: E402
from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor  # noqa: E402
from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer  # noqa: E402
from freepose_amd.src.pipeline.utils import Proposals  # noqa: E402


FOLLOW = True


def main():
    from scipy.spatial.transform import Rotation as Rot
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    fe = DINOv2FeatureExtractor("dinov2_vitl14_reg", seed=0)
    mv, mf, mc = bench.synthetic_mesh(4)
    est = DinoOnlinePoseEstimator(n_coarse_poses=600, n_fine_poses=20000, cache_size=4, cache_dir="/tmp/fp_store_curve", feature_extractor=fe)
    r600 = MeshRenderer(600)
    renders = r600.render(TriMesh(mv, mf, mc), scale=0.25)
    crops, _, _ = MeshRenderer.generate_proposals(renders)
    template = {"templates": crops.float(), "depths": renders.depth, "model_name": "curve_mesh", "intrinsic": torch.tensor([[600, 0, 210], [0, 600, 210], [0, 0, 1]])}
    est.coarse_estimator._get_template_features(template)
    H, W, scale = 720, 1280, 0.10
    f = float(np.sqrt(H ** 2 + W ** 2))
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]])
    dm = ops.Mesh(mv, mf, mc)
    R0 = np.array(est.coarse_estimator.mesh_poses[37])[:3, :3]
    ax = np.array([0.2, 1.0, 0.1])
    ax /= np.linalg.norm(ax)
    rng = np.random.Generator(np.random.PCG64(3))
    print(f"one object, {n_frames} frames per clip, ViT-L/14-reg @420^2, 20 000-rotation fine grid, 15 deg neighbourhood")
    for deg in (0.0, 0.5, 1.5, 3.0, 6.0, 12.0):
        frames, gts = [], []
        for fr in range(n_frames):
            P = np.eye(4)
            P[:3, :3] = Rot.from_rotvec(np.deg2rad(deg * fr) * ax).as_matrix() @ R0
            P[:3, 3] = [0.05, -0.02, 0.9]
            rgb, depth = ops.rasterize(dm, torch.from_numpy(P[None].astype(np.float32)), scale, f, f, W / 2.0, H / 2.0, W, H)
            m = (depth[0] > 0).cpu().numpy()
            img = rng.integers(0, 50, size=(H, W, 3), dtype=np.uint8)
            img[m] = rgb[0].cpu().numpy()[m]
            ys, xs = np.nonzero(m)
            box = torch.tensor([[int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())]])
            pr = Proposals(img, {"boxes": box, "masks": torch.from_numpy(m[None])}, 420, bbox_extend=0.05)
            frames.append((pr.proposals[0], pr.proposals_masks[0], box[0]))
            gts.append(P)
        res = {}
        for cap in (0, 768):
            est.hypothesis_cache = cap
            mesh = TriMesh(mv, mf, mc)                       # a new mesh object: a new (empty) store
            crops_seen = [0]
            inner = est.feature_extractor

            class Counting(torch.nn.Module):
                def forward(self, x, **kw):
                    crops_seen[0] += x.shape[0]
                    return inner(x, **kw)
            est.feature_extractor = Counting()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            prev, poses = None, []
            for fr, (c, cm, b) in enumerate(frames):
                if prev is None:
                    out = est(c, cm, template, mesh, K, b, scale, prev_pose=None, neighborhood=15, layer=22, batch_size=128)
                else:
                    out = est.forward_fine_many([dict(proposal=c, proposal_mask=cm, template_dict=template, mesh=mesh, K=K, bbox=b, est_scale=scale, prev_pose=prev)],
                                                neighborhood=15, layer=22)[0]
                # the weights are random-init here (no checkpoint offline), so the estimate does not follow the object; the chain is driven by the
                # DRAWN pose of this frame instead — the neighbourhood then moves through the grid exactly as fast as the object turns, which is
                # what a working tracker's would
                prev = gts[fr] if FOLLOW else out["TCO"][0]
                poses.append(out["TCO"][0].copy())
            torch.cuda.synchronize()
            est.feature_extractor = inner
            res[cap] = ((time.perf_counter() - t0) / n_frames * 1e3, crops_seen[0] / n_frames, poses)
        same = all(np.array_equal(a, b) for a, b in zip(res[0][2], res[768][2]))
        print(f"  {deg:5.1f} deg / frame: recompute {res[0][0]:6.2f} ms per frame ({res[0][1]:5.1f} crops through the ViT per frame) | store {res[768][0]:6.2f} ms "
              f"({res[768][1]:5.1f} crops)   x{res[0][0] / res[768][0]:.2f}   poses identical: {same}")


if __name__ == "__main__":
    main()
