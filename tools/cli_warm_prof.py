"""Where a warm `scripts.dino_inference` image goes (BASELINE config 3 through the CLI loop, template + feature stores resident): wall
time per image with and without the frame read-ahead, and the host profile of the loop.    python tools/cli_warm_prof.py [images] [meshes]"""
import cProfile
import io
import json
import os
import pstats
import shutil
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from freepose_amd import ops  # noqa: E402


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n_meshes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    T = 600
    from tests import _synth_scene as sc
    from scripts import dino_inference
    from freepose_amd.src.dataloader.bop import BOPDataset
    from freepose_amd.src.dataloader.template import WebTemplateDataset
    from freepose_amd.src.pipeline.estimators.pose_estimator import DinoPoseEstimator
    from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    vit = ops.ViT("dinov2_vitl14_reg", seed=0)
    root, names = bench.make_cli_workspace(n_meshes, T)
    cwd = os.getcwd()
    try:
        os.chdir(root)
        fe = DINOv2FeatureExtractor.__new__(DINOv2FeatureExtractor)
        torch.nn.Module.__init__(fe)
        fe.model_name, fe.model, fe.num_register_tokens = "dinov2_vitl14_reg", vit, vit.n_reg
        frames, props, gts, K = sc.draw_frames(root, n_img, T)
        for fr in range(n_img):
            for o, e in enumerate(props[fr]):
                e["mesh"] = names[(2 * fr + o) % n_meshes]
        sc.write_bop(root, "synth", frames, props, K)
        flat = json.loads((root / "data" / "results" / "synth" / "props.json").read_text())
        shards = root / "data" / "datasets" / "objaverse_shards"
        dataset = BOPDataset("data/datasets/synth/", "test")
        templates = WebTemplateDataset(shards.as_posix(), "data/mesh_cache.csv", bbox_extend=0.05, n_views=T, cache_meshes=n_meshes)
        model = DinoPoseEstimator(n_poses=T, cache_size=n_meshes, cache_dir=root / "cache", feature_extractor=fe)
        rows = {}
        if os.environ.get("COLD_PROFILE") == "1":          # the FIRST pass (every mesh new: decode + 600 ViT forwards + feature store fill) under the profiler
            a = dino_inference.build_parser().parse_args(["--dataset", "synth", "--proposals", "props.json", "--n_views", str(T)])
            pr = cProfile.Profile()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pr.enable()
            dino_inference.process_images(model, templates, dataset, flat, list(range(min(len(dataset), max(2, n_meshes // 2)))), a)
            torch.cuda.synchronize()
            pr.disable()
            print(f"cold pass: {n_meshes} new meshes in {(time.perf_counter() - t0) * 1e3:.0f} ms")
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
            print("\n".join(l[:170] for l in s.getvalue().splitlines()[:70]))
        for ahead, win in ((2, 1), (0, 1), (2, 4), (2, 8), (2, 1), (0, 1), (2, 4), (2, 8)):
            a = dino_inference.build_parser().parse_args(["--dataset", "synth", "--proposals", "props.json", "--n_views", str(T)])
            a.read_ahead, a.image_window = ahead, win
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = dino_inference.process_images(model, templates, dataset, flat, list(range(len(dataset))), a)
            torch.cuda.synchronize()
            sec = time.perf_counter() - t0
            rows[ahead] = r if win == 1 else rows.get(ahead, r)
            assert json.dumps(r, default=str) == json.dumps(rows.get(2, r), default=str), "the window changed the rows"
            print(f"read_ahead={ahead} image_window={win}: {len(r)} proposals of {n_img} images in {sec * 1e3:.1f} ms = {sec / n_img * 1e3:.2f} ms per image, {len(r) / sec:.0f} proposals/s")
        assert json.dumps(rows[2], default=str) == json.dumps(rows[0], default=str), "read-ahead changed the rows"
        a.read_ahead, a.image_window = 0, 1
        pr = cProfile.Profile()
        pr.enable()
        dino_inference.process_images(model, templates, dataset, flat, list(range(len(dataset))), a)
        torch.cuda.synchronize()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
        print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
    finally:
        os.chdir(cwd)
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
