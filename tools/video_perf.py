"""Secondary measurement (development): the per-frame, per-object step of the video path (BASELINE config 4 / SURVEY §3.4):
DinoOnlinePoseEstimator.forward_fine — geodesic neighbourhood of the previous pose on the 20 000-rotation grid, render the
neighbours, crop, ViT-L, patchwise score, arg-max, metric pose — chained over frames through prev_pose."""
import sys
import tempfile
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


def main():
    fe = DINOv2FeatureExtractor("dinov2_vitl14_reg", seed=0)
    with tempfile.TemporaryDirectory() as td:
        import os
        # VIDEO_STORE=0: every hypothesis recomputed per frame (the reference's step); default: the estimator's hypothesis store — with this
        # tool's STATIC query the pose settles and every later frame finds its whole neighbourhood in the store (the store's best case)
        est = DinoOnlinePoseEstimator(n_coarse_poses=8, n_fine_poses=20000, cache_size=0, cache_dir=Path(td) / "c", feature_extractor=fe,
                                      hypothesis_cache=int(os.environ.get("VIDEO_STORE", "768")))
        v, f, c = bench.synthetic_mesh(6)
        mesh = TriMesh(v, f, c)
        K = np.array([[600.0, 0, 210], [0, 600.0, 210], [0, 0, 1]])
        pose = est.fine_mesh_poses[7777]
        render = est.renderer.render_from_poses(mesh, [pose], scale=0.25)
        crops, _, masks, ext = MeshRenderer.generate_proposals(render, return_extents=True)
        e = ext[0].cpu().numpy()
        bbox = torch.tensor([int(e[0]), int(e[1]), int(e[2]), int(e[3])])
        prev = est.fine_mesh_poses[np.argsort(DinoOnlinePoseEstimator.geodesic_distance(est.fine_mesh_poses[:, :3, :3], pose))[3]]
        import os
        objs = [int(x) for x in os.environ.get("VIDEO_OBJECTS", "1").split(",")]
        for n_obj in objs:
            meshes = [mesh] + [TriMesh(v, f, c) for _ in range(n_obj - 1)]
            for neighborhood in ((15, 25) if n_obj == 1 and not os.environ.get("VIDEO_ONLY15") else (15,)):
                n_nb = len(ops.geodesic_select(est._fine_rots_dev, np.asarray(prev)[:3, :3], float(neighborhood)))

                def step(ps):
                    items = [dict(proposal=crops[0].float(), proposal_mask=masks[0], template_dict=None, mesh=meshes[o], K=K, bbox=bbox,
                                  est_scale=0.25, prev_pose=ps[o]) for o in range(n_obj)]
                    return [r["TCO"][0] for r in est.forward_fine_many(items, neighborhood=neighborhood)]
                ps = [prev] * n_obj
                for _ in range(3):
                    ps = step(ps)
                torch.cuda.synchronize()
                n = 20
                t0 = time.perf_counter()
                ps = [prev] * n_obj
                for _ in range(n):
                    ps = step(ps)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n / n_obj
                print(f"video step (ViT-L @420^2 crops, 81 920-triangle mesh), {n_obj} object(s) per frame, neighbourhood {neighborhood} deg = "
                      f"{n_nb} hypotheses at the start pose: {dt * 1e3:.2f} ms per (frame, object) = {1 / dt:.1f} frame-objects/s per GPU", flush=True)


if __name__ == "__main__":
    main()
