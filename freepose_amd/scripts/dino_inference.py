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


def window_rows(model, templates, images, layer, batch_size, bbox_extend, t_scale=1000.0, time_value=0.2, upcoming=()):
    """pose rows for the proposals of SEVERAL images (the per-proposal hot loop, reference :104-127).  `images`: dicts with image, K,
    scene_id, frame_id, props (the image's proposals-JSON entries) and scales.  The query crops of all of them share ONE ViT call — a
    proposal depends on its image only, and two crops alone are the ViT's least efficient batch (1.6 ms per crop against ~0.7 ms in a
    batch of 8) — and ONE batched estimator step with one device -> host copy; per-crop results are those of the per-image call, bit for
    bit.  `upcoming`: mesh names of the proposals that follow (the proposals JSON knows them): their templates are read and decoded in
    the background while these proposals are scored (WebTemplateDataset.prefetch)."""
    flat = []                                           # (image index, proposal index, crop, box)
    for n, im in enumerate(images):
        sp = im["props"]
        masks = torch.from_numpy(np.stack([rle_to_mask(p["segmentation"]) for p in sp]))
        boxes = torch.from_numpy(np.stack([np.array(p["bbox"]) for p in sp]))
        boxes[:, 2:] += boxes[:, :2]                     # xywh -> xyxy (:102)
        proposals = Proposals(im["image"], {"boxes": boxes, "masks": masks}, 420, bbox_extend=bbox_extend)
        flat += [(n, i, crop, boxes[i]) for i, crop in enumerate(proposals.proposals)]
    if not flat:
        return []
    # one ViT call for all proposals (the reference runs one B = 1 forward per proposal, :112-114)
    feats = model.feature_extractor(torch.stack([torch.as_tensor(c) for _, _, c, _ in flat]), layer=layer, feature_type="patch")
    ahead = [images[n]["props"][i]["mesh"] for n, i, _, _ in flat] + list(upcoming)

    def template_of(j):
        def load():
            if hasattr(templates, "prefetch_by_name"):
                for nxt in ahead[j + 1:j + 1 + PREFETCH_DEPTH]:
                    templates.prefetch_by_name(nxt)
            return templates.get_template_by_name(ahead[j])
        return load
    # ONE batched estimator step: each proposal's kernels are enqueued as its templates arrive, the scores / indices / extents of all
    # of them come back in one device -> host copy (DinoPoseEstimator.forward_many == forward per item)
    items = [dict(proposal=crop, template_dict=template_of(j), K=images[n]["K"], bbox=box, est_scale=images[n]["scales"][i], query_feat=feats[j:j + 1])
             for j, (n, i, crop, box) in enumerate(flat)]
    outs = model.forward_many(items, layer=layer, batch_size=batch_size) if hasattr(model, "forward_many") else \
        [model(it["proposal"], it["template_dict"](), it["K"], it["bbox"], it["est_scale"], layer=layer, batch_size=batch_size, query_feat=it["query_feat"])
         for it in items]
    rows = []
    for (n, i, _, _), out in zip(flat, outs):
        im = images[n]
        rows.append(pose_row(im["scene_id"], im["frame_id"], im["props"][i]["mesh"], out["scores"][0], out["TCO"][0], out["bbox"].cpu().numpy(),
                             im["scales"][i], t_scale=t_scale, time_value=time_value))
    return rows


def proposal_rows(model, templates, image, K, scene_id, frame_id, scene_props, scales, layer, batch_size, bbox_extend,
                  t_scale=1000.0, time_value=0.2, upcoming=()):
    """pose rows for the proposals of ONE image: window_rows() of a window of one"""
    return window_rows(model, templates, [dict(image=image, K=K, scene_id=scene_id, frame_id=frame_id, props=scene_props, scales=scales)],
                       layer, batch_size, bbox_extend, t_scale, time_value, upcoming)


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
    with ThreadPoolExecutor(max_workers=min(4, depth), thread_name_prefix="fp-frames") as pool:     # (a 640 x 480 PNG takes ~4 ms to decode)
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
    window = max(1, getattr(args, "image_window", 8))
    pending = []

    def flush(upcoming):
        nonlocal rows, pending
        if pending:
            rows += window_rows(model, templates, pending, args.layer, args.batch_size, args.bbox_extend, upcoming=upcoming)
        pending = []
    for n, entry in enumerate(read_ahead(dataset, images, max(getattr(args, "read_ahead", 2), window if getattr(args, "read_ahead", 2) > 0 else 0))):
        sid, fid = int(entry["scene_id"]), int(entry["frame_id"])
        sp = by_image.get((sid, fid), [])
        if not sp:
            continue
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
        pending.append(dict(image=entry["image"], K=entry["intrinsic"], scene_id=sid, frame_id=fid, props=sp, scales=scales))
        if len(pending) == window:       # the proposals of `window` images: one ViT call, one estimator step (window_rows)
            flush([p["mesh"] for k in keys[n + 1:n + 2] if k is not None for p in by_image.get(k, [])][:PREFETCH_DEPTH])
    flush(())
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
    ap.add_argument("--image_window", type=int, default=8)                   # not in the reference: images whose proposals share one ViT call and one estimator step (1 = per image, same CSV)
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
