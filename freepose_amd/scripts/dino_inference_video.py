"""python -m scripts.dino_inference_video --video <name> --proposals <props>.json

Drop-in for the reference CLI (scripts/dino_inference_video.py:43-182, flags :230-241): per frame and object, online
render-and-compare with `prev_pose` chaining; pose CSV out (t in metres, scene_id 0, time -1).

Multi-GPU (new): frames of one object are sequentially dependent through prev_pose (reference :122,155-156), so a
torch.distributed.run launch shards OBJECTS across ranks — exact w.r.t. the reference — and all-gathers the 19-float pose rows.
`--no_rescore` (coarse pose per frame, frames independent) shards FRAMES instead; the reference's own --no_rescore reads
a key the coarse estimator never returns, here it simply emits the coarse top-1 pose.
`--frame_chunks` (new, DEVIATES from the reference): each rank tracks a contiguous stretch of the clip and re-initialises
with a coarse estimate at the head of its stretch (SURVEY §8e option 4) — this is what lets a single-object clip use all
8 GPUs; poses on chunk-initial frames (and possibly after) differ from a sequential run, so the CSV name gets `_chunked<N>`.
`--n_views`, `--model`, `--n_fine_poses` (new, default to the reference's constants 600 / dinov2_vitl14_reg / 20000).
"""
from __future__ import annotations

import argparse
import functools
import json
import os
from itertools import takewhile
from pathlib import Path

import numpy as np
import pandas as pd
import torch
from PIL import Image

from freepose_amd import ops, parallel
from freepose_amd.mesh_io import load_obj
from freepose_amd.src.dataloader.template import WebTemplateDataset
from freepose_amd.src.pipeline.estimators.online_pose_estimator import DinoOnlinePoseEstimator
from freepose_amd.src.pipeline.estimators.pose_estimator import DinoPoseEstimator
from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
from freepose_amd.src.pipeline.utils import Proposals, rle_to_mask
from freepose_amd.scripts.dino_inference import CSV_COLUMNS, pose_row, read_ahead


def guessed_intrinsics(h: int, w: int) -> np.ndarray:
    f = np.sqrt(h ** 2 + w ** 2)                     # reference :115-118
    return np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1]]).astype(float)


def track_objects(model, templates, frames, props, meshes, mesh_ids, scales, K, obj_ids, args, rescoring=True,
                  frame_ids=None):
    """pose rows [(frame, obj, score, TCO, bbox)] for the given objects over the given frames (all by default).  With
    rescoring the first frame visited starts from a coarse estimate (prev_pose None), every later one from its predecessor."""
    prev = {o: None for o in obj_ids}
    rows = []
    if hasattr(templates, "prefetch_by_name"):       # the clip's meshes are known up front: read / decode them behind the first frame's work
        for o in list(obj_ids)[1:3]:
            templates.prefetch_by_name(mesh_ids[o])
    class _FrameInputs:          # host side of a frame (JPEG decode, RLE masks, boxes): nothing in it depends on the previous pose
        def __getitem__(self, f):
            sp = props[f]
            img = np.asarray(Image.open(frames[f]).convert("RGB"), dtype=np.uint8)
            masks = torch.from_numpy(np.stack([rle_to_mask(p["segmentation"]) for p in sp]))
            boxes = torch.from_numpy(np.stack([np.array(p["bbox"]) for p in sp]))
            boxes[:, 2:] += boxes[:, :2]
            return f, img, masks, boxes
    order = list(range(len(frames)) if frame_ids is None else frame_ids)
    depth = getattr(args, "read_ahead", 2)
    window = max(1, getattr(args, "query_window", 8))

    def windows():
        """lists of up to `window` consecutive frames, each (f, boxes, Proposals): the next frames are decoded on a background thread
        while this window's render-and-compare steps run (read_ahead)"""
        chunk = []
        for f, img, masks, boxes in read_ahead(_FrameInputs(), order, max(depth, window if depth > 0 else 0)):
            chunk.append((f, boxes, Proposals(img, {"boxes": boxes, "masks": masks}, 420, bbox_extend=args.bbox_extend)))
            if len(chunk) == window:
                yield chunk
                chunk = []
        if chunk:
            yield chunk

    for chunk in windows():
        # QUERY features of the whole window in ONE ViT call: a query crop depends on its frame only, not on the pose chain, and one
        # crop alone is the ViT's least efficient batch (2.5 ms against ~0.7 ms per crop in a batch of 8); same bits per crop as the
        # per-frame call.  The very first frame visited goes through the coarse estimator, which encodes its own query.
        qn = {}
        if window > 1:
            with torch.inference_mode():
                todo = [(f, o) for f, _, _ in chunk for o in obj_ids if not (rescoring and f == order[0])]
                if todo:
                    by_f = {f: pr for f, _, pr in chunk}
                    crops = torch.stack([torch.as_tensor(by_f[f].proposals[o]) for f, o in todo]).to("cuda", torch.bfloat16)
                    feats = model.feature_extractor(crops, layer=args.layer, feature_type="patch")
                    if rescoring:                  # the fine step scores against the NORMALISED query (online_pose_estimator.py:58-60); the
                        feats = ops.l2_normalize(feats)        # coarse estimator normalises its raw query itself (pose_estimator.py:87)
                    qn = {fo: feats[i:i + 1] for i, fo in enumerate(todo)}
        for f, boxes, proposals in chunk:
            outs = {}
            with torch.inference_mode():
                if rescoring:
                    # objects that already have a pose: ONE batched render-and-compare step for the whole frame (the reference visits
                    # them one by one, :124-156; they are independent, and the batched ViT call returns the same bits per crop)
                    many = [o for o in obj_ids if prev[o] is not None]
                    if many:
                        items = [dict(proposal=proposals.proposals[o], proposal_mask=proposals.proposals_masks[o],
                                      template_dict=templates.get_template_by_name(mesh_ids[o]), mesh=meshes[o], K=K, bbox=boxes[o],
                                      est_scale=scales[o], prev_pose=prev[o], query_feat=qn.get((f, o))) for o in many]
                        for o, out in zip(many, model.forward_fine_many(items, neighborhood=15, layer=args.layer)):
                            outs[o] = out
                    for o in obj_ids:
                        if o not in outs:     # first frame visited: coarse estimate, then the fine step (prev_pose None)
                            outs[o] = model(proposals.proposals[o], proposals.proposals_masks[o], templates.get_template_by_name(mesh_ids[o]),
                                            meshes[o], K, boxes[o], scales[o], prev_pose=None, neighborhood=15, layer=args.layer,
                                            batch_size=args.batch_size)
                        prev[o] = outs[o]["TCO"][0]
                elif window > 1:       # coarse estimate per frame (frames independent): the frame's objects in one estimator step
                    items = [dict(proposal=proposals.proposals[o], template_dict=templates.get_template_by_name(mesh_ids[o]), K=K, bbox=boxes[o],
                                  est_scale=scales[o], query_feat=qn[(f, o)]) for o in obj_ids]
                    for o, out in zip(obj_ids, model.forward_many(items, layer=args.layer, batch_size=args.batch_size)):
                        outs[o] = out
                else:
                    for o in obj_ids:
                        outs[o] = model(proposals.proposals[o], templates.get_template_by_name(mesh_ids[o]), K, boxes[o], scales[o],
                                        layer=args.layer, batch_size=args.batch_size)
            for o in obj_ids:
                rows.append((f, o, float(outs[o]["scores"][0]), outs[o]["TCO"][0], boxes[o].numpy()))
    return rows


def main(args):
    rank, world, _ = parallel.init_from_env()
    if world > 1:
        parallel.announce("dist")          # backend, RCCL version, device + PCI bus id of every rank
    video_dir = (Path("data") / "datasets" / "videos" / args.video).resolve()
    frames = sorted(p for p in video_dir.iterdir() if p.suffix.lower() in (".jpg", ".jpeg"))
    results_dir = (Path("data") / "results" / "videos" / args.video).resolve()
    chunked = bool(args.frame_chunks) and world > 1 and not args.no_rescore
    out_csv = results_dir / args.proposals.replace(
        ".json", f"_dinopose_layer_{args.layer}_bbext_{args.bbox_extend}_depth_{args.depth_method}"
                 + (f"_chunked{world}" if chunked else "") + ".csv")

    templates = WebTemplateDataset("data/datasets/objaverse_shards", "data/mesh_cache.csv", bbox_extend=args.bbox_extend,
                                   n_views=args.n_views)
    templates.get_template_by_name = functools.lru_cache(maxsize=args.template_cache_size)(templates.get_template_by_name)
    cache_dir = Path("data") / f"cache_{os.environ.get('SLURM_JOB_ID', 0)}_{args.video}_r{rank}"
    extractor = DINOv2FeatureExtractor(args.model, allow_random_weights=args.allow_random_weights or None)
    if args.no_rescore:
        model = DinoPoseEstimator(n_poses=args.n_views, cache_size=args.cache_size, save_all=args.save_all_cache,
                                  cache_dir=cache_dir, feature_extractor=extractor)
    else:
        model = DinoOnlinePoseEstimator(n_coarse_poses=args.n_views, n_fine_poses=args.n_fine_poses, cache_size=args.cache_size,
                                        save_all=args.save_all_cache, cache_dir=cache_dir, feature_extractor=extractor,
                                        hypothesis_cache=args.hypothesis_cache, hypothesis_meshes=args.hypothesis_meshes)

    props = json.loads((results_dir / args.proposals).read_text())
    n_objects = len(list(takewhile(lambda x: x["image_id"] == 0, props)))
    n_frames = len(frames)
    assert n_objects * n_frames == len(props)
    props = [props[i:i + n_objects] for i in range(0, len(props), n_objects)]
    if args.depth_method.startswith("const-"):
        scales = [float(args.depth_method.split("-")[1])] * n_objects
    elif args.depth_method == "zoedepth":
        scales = [props[0][o]["scale"] for o in range(n_objects)]
        for o in range(n_objects):
            assert all(props[f][o]["scale"] == scales[o] for f in range(n_frames)), f"Object {o} has different scales"
    else:
        raise NotImplementedError()
    mesh_ids, meshes = [], []
    for o in range(n_objects):
        mid = props[0][o]["mesh"]
        assert all(props[f][o]["mesh"] == mid for f in range(n_frames)), f"Object {o} has different meshes"
        meshes.append(load_obj(Path("data").resolve() / "mesh_cache" / mid / f"{mid}.obj"))
        mesh_ids.append(mid)
    h, w = np.asarray(Image.open(frames[0])).shape[:2]
    K = guessed_intrinsics(h, w)

    if args.no_rescore:   # frames independent -> shard frames
        rows = track_objects(model, templates, frames, props, meshes, mesh_ids, scales, K, list(range(n_objects)), args,
                             rescoring=False, frame_ids=parallel.shard_items(n_frames, rank, world))
    elif chunked:         # DEVIATES: every rank tracks its own stretch of the clip, coarse re-init at the stretch head
        if rank == 0:
            print(f"[dino_inference_video] --frame_chunks: {world} chunks with a coarse re-initialisation each; poses deviate "
                  "from a sequential run on chunk-initial frames (SURVEY 8e option 4)", flush=True)
        rows = track_objects(model, templates, frames, props, meshes, mesh_ids, scales, K, list(range(n_objects)), args,
                             frame_ids=parallel.shard_chunk(n_frames, rank, world))
    else:                 # prev_pose chains frames -> shard objects
        rows = track_objects(model, templates, frames, props, meshes, mesh_ids, scales, K,
                             parallel.shard_items(n_objects, rank, world), args)
    packed = torch.tensor([[f, o, s, *T[:3, :3].flatten(), *T[:3, 3], *b] for f, o, s, T, b in rows], dtype=torch.float64)
    if world > 1:
        packed = parallel.all_gather_rows(packed.reshape(-1, 19).cuda()).cpu()
    if rank == 0:
        packed = packed[np.lexsort((packed[:, 1].numpy(), packed[:, 0].numpy()))] if len(packed) else packed
        recs = []
        for r in packed.numpy():
            f, o = int(r[0]), int(r[1])
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = r[3:12].reshape(3, 3), r[12:15]
            recs.append(pose_row(0, f, mesh_ids[o], r[2], T, r[15:19], scales[o], t_scale=1, time_value=-1))
        results_dir.mkdir(parents=True, exist_ok=True)
        pd.DataFrame(recs, columns=CSV_COLUMNS).to_csv(out_csv, index=False, header=True)


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--video", type=str, required=True)
    ap.add_argument("--proposals", type=str, required=True)
    ap.add_argument("--layer", type=int, default=22)
    ap.add_argument("--depth_method", type=str, default="zoedepth")
    ap.add_argument("--bbox_extend", type=float, default=0.05)
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--template_cache_size", type=int, default=21)
    ap.add_argument("--viz", action="store_true", help="accepted for CLI parity; visualisation is out of scope")
    ap.add_argument("--no_rescore", action="store_true")
    ap.add_argument("--cache_size", type=int, default=50)
    ap.add_argument("--save_all_cache", action="store_true")
    # not in the reference (defaults reproduce it): template views per mesh = coarse hypotheses, backbone, fine grid size,
    # and the deviating frame-chunk mode
    ap.add_argument("--n_views", type=int, default=600)
    ap.add_argument("--model", type=str, default="dinov2_vitl14_reg")
    ap.add_argument("--n_fine_poses", type=int, default=20000)
    ap.add_argument("--frame_chunks", action="store_true")
    ap.add_argument("--allow_random_weights", action="store_true")           # run without the DINOv2 checkpoint (tests, benches)
    ap.add_argument("--gpus", type=int, default=1)                           # self-launch N ranks, one per GPU (RCCL)
    ap.add_argument("--read_ahead", type=int, default=2)                     # frames decoded ahead on a thread (0 = the sequential loop)
    ap.add_argument("--hypothesis_cache", type=int, default=768)             # fine-grid hypotheses kept per mesh between frames (0 = recompute all, same CSV)
    ap.add_argument("--hypothesis_meshes", type=int, default=8)              # meshes that keep such a store between frames (a frame's own objects always do)
    ap.add_argument("--query_window", type=int, default=8)                   # frames whose query crops share one ViT call (1 = one call per frame, same CSV)
    return ap


def run(argv=None):
    import sys
    args = build_parser().parse_args(argv)
    parallel.self_launch(args.gpus, ["-m", "scripts.dino_inference_video"], sys.argv[1:] if argv is None else list(argv))
    return main(args)


if __name__ == "__main__":
    run()
