"""python -m scripts.dino_inference --dataset ycbv --proposals <props>.json

Drop-in for the reference CLI (scripts/dino_inference.py:22-130): proposals JSON in, pose CSV out
(`scene_id,im_id,obj_id,score,R,t,bbox_visib,scale,time`; t in millimetres :124; same output path :38-40).  Same flags.
Images are sliced by SLURM_ARRAY_TASK_ID (30 per task, :37,51-54) and — new — additionally round-robin over the ranks of
a torch.distributed.run launch; each rank writes its own CSV like the reference's per-task files (merge_results.py
concatenates them)."""
from __future__ import annotations

import argparse
import json
import os
from pathlib import Path

import numpy as np
import pandas as pd
import torch

from freepose_amd import parallel
from freepose_amd.src.dataloader.bop import BOPDataset
from freepose_amd.src.dataloader.template import WebTemplateDataset
from freepose_amd.src.pipeline.estimators.pose_estimator import DinoPoseEstimator
from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
from freepose_amd.src.pipeline.utils import Proposals, rle_to_mask

CSV_COLUMNS = ["scene_id", "im_id", "obj_id", "score", "R", "t", "bbox_visib", "scale", "time"]


def pose_row(scene_id, im_id, obj_id, score, TCO, bbox_xyxy, scale, t_scale=1000.0, time_value=0.2):
    """one CSV record, formatted like the reference (image script :113-127: t in millimetres, time 0.2; video script
    scripts/dino_inference_video.py:160-176: t_scale 1, time -1).  Pinned by tests/golden/csv_rows.npz."""
    TCO = np.asarray(TCO, dtype=np.float64)
    b = [int(x) for x in np.asarray(bbox_xyxy).reshape(-1)[:4]]
    return {"scene_id": int(scene_id), "im_id": int(im_id), "obj_id": obj_id, "score": np.float32(score),
            "R": " ".join(str(x) for x in TCO[:3, :3].flatten().tolist()),
            "t": " ".join(str(x * t_scale) if t_scale != 1 else str(x) for x in TCO[:3, 3].tolist()),
            "bbox_visib": " ".join(str(x) for x in [b[0], b[1], b[2] - b[0], b[3] - b[1]]),
            "scale": scale, "time": time_value}


PREFETCH_DEPTH = 2     # template meshes read / decoded ahead of the proposal being scored (530 MB of device memory each while they wait)


def proposal_rows(model, templates, image, K, scene_id, frame_id, scene_props, scales, layer, batch_size, bbox_extend,
                  t_scale=1000.0, time_value=0.2, upcoming=()):
    """pose rows for the proposals of ONE image (the per-proposal hot loop, reference :104-127).  `upcoming`: the mesh names of the
    proposals that follow this image (the proposals JSON knows them): their templates are read and decoded in the background while
    this image's proposals are scored (WebTemplateDataset.prefetch)."""
    masks = torch.from_numpy(np.stack([rle_to_mask(p["segmentation"]) for p in scene_props]))
    boxes = torch.from_numpy(np.stack([np.array(p["bbox"]) for p in scene_props]))
    boxes[:, 2:] += boxes[:, :2]                         # xywh -> xyxy (:102)
    proposals = Proposals(image, {"boxes": boxes, "masks": masks}, 420, bbox_extend=bbox_extend)
    rows = []
    # one ViT call for all proposals of the image (the reference runs one B = 1 forward per proposal, :112-114)
    crops = list(proposals.proposals)
    feats = model.feature_extractor(torch.stack([torch.as_tensor(c) for c in crops]), layer=layer, feature_type="patch") if crops else None
    ahead = [p["mesh"] for p in scene_props] + list(upcoming)

    def template_of(i):
        def load():
            if hasattr(templates, "prefetch_by_name"):
                for nxt in ahead[i + 1:i + 1 + PREFETCH_DEPTH]:
                    templates.prefetch_by_name(nxt)
            return templates.get_template_by_name(scene_props[i]["mesh"])
        return load
    # the image's proposals go through ONE batched estimator step: each proposal's kernels are enqueued as its templates arrive, the
    # scores / indices / extents of all of them come back in one device -> host copy (DinoPoseEstimator.forward_many == forward per item)
    items = [dict(proposal=prop, template_dict=template_of(i), K=K, bbox=boxes[i], est_scale=scales[i], query_feat=feats[i:i + 1])
             for i, prop in enumerate(crops)]
    outs = model.forward_many(items, layer=layer, batch_size=batch_size) if hasattr(model, "forward_many") else \
        [model(it["proposal"], it["template_dict"](), K, it["bbox"], it["est_scale"], layer=layer, batch_size=batch_size, query_feat=it["query_feat"])
         for it in items]
    for i, out in enumerate(outs):
        mesh = scene_props[i]["mesh"]
        rows.append(pose_row(scene_id, frame_id, mesh, out["scores"][0], out["TCO"][0], out["bbox"].cpu().numpy(), scales[i],
                             t_scale=t_scale, time_value=time_value))
    return rows


def read_ahead(dataset, images, depth=2):
    """dataset[idx] for idx in images, in order, with the next `depth` entries being read and decoded (PIL releases the GIL) on a
    background thread while the caller works on the current one — a 640 x 480 BOP frame decodes in about the time the GPU needs for
    its proposals"""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    images = list(images)
    if depth <= 0 or len(images) < 2:
        for idx in images:
            yield dataset[idx]
        return
    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="fp-frames") as pool:
        pending = deque(pool.submit(dataset.__getitem__, idx) for idx in images[:depth])
        for n in range(len(images)):
            entry = pending.popleft().result()
            if n + depth < len(images):
                pending.append(pool.submit(dataset.__getitem__, images[n + depth]))
            yield entry


def process_images(model, templates, dataset, props, images, args):
    """the per-image loop of the driver (reference :69-127): pose rows of the given dataset entries"""
    rows = []
    by_image = {}
    for p in props:
        by_image.setdefault((p["scene_id"], p["image_id"]), []).append(p)
    keys = [dataset.frame_key(idx) if hasattr(dataset, "frame_key") else None for idx in images]
    for n, entry in enumerate(read_ahead(dataset, images, getattr(args, "read_ahead", 2))):
        sid, fid = int(entry["scene_id"]), int(entry["frame_id"])
        sp = by_image.get((sid, fid), [])
        if not sp:
            continue
        upcoming = [p["mesh"] for k in keys[n + 1:n + 2] if k is not None for p in by_image.get(k, [])][:PREFETCH_DEPTH]
        if args.depth_method == "zoedepth":
            scales = [float(np.clip(p["scale"], a_min=0.01, a_max=None)) for p in sp]
        elif args.depth_method.startswith("const-"):
            scales = [float(args.depth_method.split("-")[1])] * len(sp)
        elif args.depth_method == "depthmap":             # reference :82-85: scale from the scene depth map under each proposal mask
            from freepose_amd.src.pipeline.estimators.scale_estimators import depthmap_scale
            if "depth" not in entry:
                raise FileNotFoundError(f"--depth_method depthmap: scene {sid} frame {fid} has no depth map")
            scales = [depthmap_scale(entry["depth"], entry["intrinsic"], rle_to_mask(p["segmentation"])) for p in sp]
            for p, sc in zip(sp, scales):
                p["scale"] = sc
        else:
            raise ValueError(f"unknown --depth_method {args.depth_method} (depthmap | const-<metres> | zoedepth)")
        rows += proposal_rows(model, templates, entry["image"], entry["intrinsic"], sid, fid, sp, scales, args.layer,
                              args.batch_size, args.bbox_extend, upcoming=upcoming)
    return rows


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", type=str)
    ap.add_argument("--split", type=str, default="test")
    ap.add_argument("--proposals", type=str)
    ap.add_argument("--layer", type=int, default=22)
    ap.add_argument("--depth_method", type=str, default="zoedepth")
    ap.add_argument("--bbox_extend", type=float, default=0.05)
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--cache_size", type=int, default=50)
    ap.add_argument("--save_all_cache", action="store_true")
    ap.add_argument("--n_views", type=int, default=600)                      # not in the reference: views per mesh (600 there)
    ap.add_argument("--model", type=str, default="dinov2_vitl14_reg")        # not in the reference: backbone
    ap.add_argument("--allow_random_weights", action="store_true")           # not in the reference: run without the checkpoint
    ap.add_argument("--gpus", type=int, default=1)                           # not in the reference: self-launch N ranks, one per GPU
    ap.add_argument("--read_ahead", type=int, default=2)                     # not in the reference: frames decoded ahead on a thread (0 = the sequential loop)
    return ap


def run(argv=None):
    args = build_parser().parse_args(argv)

    import sys
    parallel.self_launch(args.gpus, ["-m", "scripts.dino_inference"], sys.argv[1:] if argv is None else list(argv))
    rank, world, _ = parallel.init_from_env()
    if world > 1:
        parallel.announce("dist")          # backend, RCCL version, device + PCI bus id of every rank
    task = int(os.getenv("SLURM_ARRAY_TASK_ID", 0))
    res_dir = Path("./data/results").resolve() / args.dataset
    out_dir = res_dir / args.proposals.replace(
        ".json", f"_dinopose_layer_{args.layer}_bbext_{args.bbox_extend}_depth_{args.depth_method}_cache_{args.cache_size}")
    out_dir.mkdir(parents=True, exist_ok=True)
    out_csv = out_dir / (f"pose_outputs_{task}.csv" if world == 1 else f"pose_outputs_{task}_r{rank}.csv")

    dataset = BOPDataset(f"data/datasets/{args.dataset}/", args.split)
    templates = WebTemplateDataset("data/datasets/objaverse_shards", "data/mesh_cache.csv", bbox_extend=args.bbox_extend,
                                   n_views=args.n_views)
    extractor = DINOv2FeatureExtractor(args.model, allow_random_weights=args.allow_random_weights or None)
    model = DinoPoseEstimator(n_poses=args.n_views, cache_size=args.cache_size, save_all=args.save_all_cache,
                              cache_dir=f"./data/cache_{task}_{args.dataset}_r{rank}", feature_extractor=extractor)
    props = json.loads((res_dir / args.proposals).read_text())

    per_task = 30
    images = list(range(task * per_task, min((task + 1) * per_task, len(dataset))))[rank::world]
    rows = process_images(model, templates, dataset, props, images, args)
    pd.DataFrame(rows, columns=CSV_COLUMNS).to_csv(out_csv, index=False, header=True)
    return out_csv


if __name__ == "__main__":
    run()
