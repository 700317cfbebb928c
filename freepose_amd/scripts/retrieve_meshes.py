"""python -m scripts.retrieve_meshes --dataset ycbv --detections dets.json           (static images)
   python -m scripts.retrieve_meshes --video clip --detections dets.json --topk 25  (video, soft vote over the frames)

The retrieval half of the reference's proposal drivers — scripts/extract_proposals_ground.py:118-163 (per proposal: masked crop ->
ViT-L layer-`L` features -> FFA or cls descriptor -> cosine top-100 over the bank -> optional per-view re-rank `--topk k` -> mesh id,
score) and scripts/extract_proposals_ground_video.py:118-197 (unmasked crops, the same retrieval per frame, then the soft vote: mean
over the frames of the dense per-object score vectors, arg-max) — with the detector / segmenter / tracker that produce the boxes
and masks (GroundingDINO, SAM 2: SURVEY §2 rows 10-11, out of scope) replaced by their OUTPUT on disk: `--detections`, a JSON list
in the proposals schema (bbox xywh, segmentation = uncompressed column-major RLE, scene_id, image_id; video files frame-
major with a constant number of objects per frame).  Writes the proposals JSON `scripts.dino_inference[_video]` consume, with `mesh`
and `score` filled in, under the reference's file name (`props-ground-box-{b}-text-{t}-{ffa|cls}-{layer}-top-{k}_{dataset}-{split}.json`
/ `..._{video}.json`; the two thresholds belong to the detector and only name the file).

Same retrieval flags as the reference (`--retrieval --filelist --topk`, defaults 0 for images and 25 for videos); `data/<retrieval>.npy`
is the bank, `data/datasets/<retrieval>/<mesh>.npy` the per-view descriptors of the re-rank (all device-resident: TemplateBank.attach_views).
Video runs under torch.distributed shard the frames; the soft vote is then a collective (parallel.soft_vote_reduce), identical on
every rank and to a single-rank run.
"""
from __future__ import annotations

import argparse
import json
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from freepose_amd import ops, parallel
from freepose_amd.retrieval import TemplateBank
from freepose_amd.scripts.dino_inference import read_ahead
from freepose_amd.src.dataloader.bop import BOPDataset
from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
from freepose_amd.src.pipeline.utils import Proposals, rle_to_mask


def detections_of(entries):
    """{'boxes': int [n,4] xyxy, 'masks': bool [n,H,W]} of one image's detection entries (bbox xywh -> xyxy like the drivers, :102)"""
    masks = torch.from_numpy(np.stack([rle_to_mask(e["segmentation"]) for e in entries]))
    boxes = torch.from_numpy(np.stack([np.array(e["bbox"]) for e in entries])).clone()
    boxes[:, 2:] += boxes[:, :2]
    return {"boxes": boxes, "masks": masks}


def describe(extractor, proposals, feature_type: str, layer: int):
    """bf16 [n, D] L2-normalised descriptors of the proposals (ground.py:123-134): cls token, or FFA = mean of the patch features under
    the crop's mask any-pooled to the 30 x 30 patch grid (cv2 INTER_AREA > 0, :127).  `proposals`: one Proposals object or a list of
    them (several images / frames: ONE ViT call for all their crops — a crop's descriptor does not depend on the crops beside it —
    and a list of descriptor tensors back, one per object)"""
    many = isinstance(proposals, (list, tuple))
    group = list(proposals) if many else [proposals]
    crops = torch.cat([torch.as_tensor(p.proposals) for p in group], dim=0)
    if feature_type == "cls":
        out = ops.l2_normalize(extractor(crops, layer=layer, feature_type="cls"))
    else:
        feats = extractor(crops, layer=layer, feature_type="patch")
        out = ops.ffa(feats, torch.cat([torch.as_tensor(p.proposals_masks) for p in group], dim=0), cell=14, normalize=True)
    if not many:
        return out
    sizes = [len(p.proposals) for p in group]
    return list(torch.split(out, sizes, dim=0))


WINDOW = 8     # images / frames whose crops share one ViT call


def load_bank(retrieval: str, filelist: str, topk: int) -> TemplateBank:
    ids = (Path("data") / filelist).read_text().splitlines()
    bank = TemplateBank(np.load(Path("data") / f"{retrieval}.npy"), ids)
    if topk:
        bank.attach_views([np.load(Path("data") / "datasets" / retrieval / f"{m}.npy") for m in ids])
    return bank


def retrieve_image(bank: TemplateBank, queries: torch.Tensor, topk: int):
    """(mesh ids, scores) per proposal: best of the coarse top-100 (topk == 0 — the reference still scans for 100 and takes the first,
    :142-145) or the candidate whose `topk` best views score highest on average (:147-160)"""
    if topk == 0:
        names, scores, _ = bank.retrieve(queries)
        return names, [float(s) for s in scores]
    names, scores, _, _ = bank.retrieve_reranked(queries, topk=topk)
    return names, [float(s) for s in scores]


def run_images(args, extractor, bank, feature_type, layer):
    dataset = BOPDataset(f"data/datasets/{args.dataset}/", args.split)
    results = Path("data/results").resolve() / args.dataset
    dets = json.loads((results / args.detections).read_text())
    by_image = {}
    for d in dets:
        by_image.setdefault((int(d["scene_id"]), int(d["image_id"])), []).append(d)
    out = []
    wanted = [idx for idx in range(len(dataset)) if by_image.get(dataset.frame_key(idx))]
    with torch.inference_mode():
        group = []

        def flush():
            for proposals, q in zip(group, describe(extractor, group, feature_type, layer) if group else []):
                proposals.meshes, proposals.scores = retrieve_image(bank, q, args.topk)
                out.extend(proposals.to_bop_dict())
            group.clear()
        for idx, entry in zip(wanted, read_ahead(dataset, wanted, WINDOW)):  # the next frames are decoded on threads meanwhile
            key = dataset.frame_key(idx)
            group.append(Proposals(entry["image"], detections_of(by_image[key]), 420, key[0], key[1], bbox_extend=0.1, mask_rgb=True))   # :121
            if len(group) == WINDOW:
                flush()
        flush()
    name = f"props-ground-box-{args.box_thresh}-text-{args.text_thresh}-{feature_type}-{layer}-top-{args.topk}_{args.dataset}-{args.split}.json"
    path = results / (args.output or name)
    path.write_text(json.dumps(out))
    return path


def run_video(args, extractor, bank, feature_type, layer):
    rank, world, _ = parallel.init_from_env()
    video_dir = (Path("data") / "datasets" / "videos" / args.video).resolve()
    frames = sorted(p for p in video_dir.iterdir() if p.suffix.lower() in (".jpg", ".jpeg"))
    results = (Path("data") / "results" / "videos" / args.video).resolve()
    dets = json.loads((results / args.detections).read_text())
    n_obj = sum(1 for d in dets if d["image_id"] == dets[0]["image_id"])
    assert n_obj > 0 and n_obj * len(frames) == len(dets), "video detections: frame-major, the same objects in every frame"
    per_frame = [dets[i:i + n_obj] for i in range(0, len(dets), n_obj)]
    mine = parallel.shard_items(len(frames), rank, world)
    queries = []
    class _Frames:
        def __getitem__(self, f):
            return np.asarray(Image.open(frames[f]).convert("RGB"), dtype=np.uint8)
    with torch.inference_mode():
        group = []
        for f, img in zip(mine, read_ahead(_Frames(), mine, WINDOW)):
            group.append(Proposals(img, detections_of(per_frame[f]), 420, 0, f, bbox_extend=0.1, mask_rgb=False))               # video :133
            if len(group) == WINDOW:
                queries.extend(describe(extractor, group, feature_type, layer))
                group = []
        if group:
            queries.extend(describe(extractor, group, feature_type, layer))
        # soft vote (:154-159,186-190): dense [N] vectors holding each frame's top-100 scores (or re-ranked means), mean over the
        # frames, arg-max per object; every frame then carries the clip's meshes and scores (:192-195)
        rows, best = bank.soft_vote(queries, k=min(100, bank.N), topk=args.topk, frame_ids=mine if world > 1 else None, n_obj=n_obj)
    meshes = [bank.mesh_ids[int(r)] for r in rows]
    scores = [float(s) for s in best]
    out = []
    for f in range(len(frames)):
        det = detections_of(per_frame[f])
        img_hw = per_frame[f][0]["segmentation"]["size"]
        p = Proposals.__new__(Proposals)          # (no crops needed for the frames another rank described: only the JSON record)
        p.masks, p.boxes, p.meshes, p.scores, p.scene_id, p.frame_id = det["masks"].bool(), det["boxes"].int(), meshes, scores, 0, f
        assert list(p.masks.shape[1:]) == list(img_hw)
        out.extend(p.to_bop_dict())
    name = f"props-ground-box-{args.box_thresh}-text-{args.text_thresh}-{feature_type}-{layer}-top-{args.topk}_{args.video}.json"
    path = results / (args.output or name)
    if rank == 0:
        path.write_text(json.dumps(out))
    return path


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", type=str)
    ap.add_argument("--split", type=str, default="test")
    ap.add_argument("--video", type=str)
    ap.add_argument("--detections", type=str, required=True)        # not in the reference: the detector / tracker output (boxes + RLE masks)
    ap.add_argument("--retrieval", type=str, default="objaverse_shards_ffa_22")
    ap.add_argument("--filelist", type=str, default="mesh_cache.txt")
    ap.add_argument("--box_thresh", type=float, default=None)       # detector thresholds: they only name the output file here
    ap.add_argument("--text_thresh", type=float, default=None)
    ap.add_argument("--topk", type=int, default=None)               # reference defaults: 0 (images), 25 (videos)
    ap.add_argument("--output", type=str, default=None)             # not in the reference: output file name
    ap.add_argument("--model", type=str, default="dinov2_vitl14_reg")
    ap.add_argument("--allow_random_weights", action="store_true")
    ap.add_argument("--gpus", type=int, default=1)
    return ap


def run(argv=None):
    import sys
    args = build_parser().parse_args(argv)
    if bool(args.dataset) == bool(args.video):
        raise SystemExit("retrieve_meshes: give exactly one of --dataset (static images) and --video")
    parallel.self_launch(args.gpus, ["-m", "scripts.retrieve_meshes"], sys.argv[1:] if argv is None else list(argv))
    video = bool(args.video)
    if args.topk is None:
        args.topk = 25 if video else 0
    if args.box_thresh is None:
        args.box_thresh = 0.2 if video else 0.3
    if args.text_thresh is None:
        args.text_thresh = 0.2 if video else 0.5
    feature_type = "ffa" if "ffa" in args.retrieval else "cls"                   # ground.py:32-33
    layer = int(args.retrieval.split("_")[-1])
    extractor = DINOv2FeatureExtractor(args.model, allow_random_weights=args.allow_random_weights or None)
    bank = load_bank(args.retrieval, args.filelist, args.topk)
    return (run_video if video else run_images)(args, extractor, bank, feature_type, layer)


if __name__ == "__main__":
    run()
