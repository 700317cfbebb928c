"""python -m scripts.extract_retrieval_features --feature ffa --layer 22 --batch_size 256

Drop-in for the reference CLI (scripts/extract_retrieval_features.py:12-75): per mesh, 600 template views -> ViT patch
features -> per-view FFA (masked mean over the any-pooled 30x30 mask) -> data/datasets/<shards>_<feature>_<layer>/<mesh>.npy
([<=600, 1024] fp32, NaN views dropped).  Same flags; the job slice comes from SLURM_ARRAY_TASK_ID like the reference,
or — new — from RANK/WORLD_SIZE when launched with torch.distributed.run (meshes round-robin over the GPUs of a node).
The 600 per-view device->host copies of the reference (:57) become one copy per mesh, and the next mesh's tar reads, PNG decode
and host->device copy run in the background while this mesh is in the ViT (`WebTemplateDataset.prefetch`; `--no_prefetch` = the
sequential loop, same files byte for byte).
"""
from __future__ import annotations

import argparse
import os
from pathlib import Path

import numpy as np
import torch

from freepose_amd import ops, parallel
from freepose_amd.src.dataloader.template import WebTemplateDataset
from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor


def mesh_descriptors(model, sample, feature: str, layer: int, batch_size: int) -> np.ndarray:
    """[<=T, D] fp32 per-view descriptors of one mesh (FFA) or [T, D] cls features."""
    templates = sample["templates"]
    ftype = "cls" if feature == "cls" else "patch"
    if hasattr(model, "forward_batched"):               # batches of about batch_size that fill whole GEMM rounds (DINOv2FeatureExtractor.forward_batched)
        feats = model.forward_batched(templates, layer=layer, feature_type=ftype, batch_size=batch_size)
    else:
        feats = torch.cat([model(templates[i:i + batch_size], layer=layer, feature_type=ftype)
                           for i in range(0, len(templates), batch_size)], dim=0)
    if feature != "ffa":
        return feats.float().cpu().numpy()
    desc = ops.ffa(feats, sample["masks"], cell=14, out_f32=True).cpu().numpy()
    keep = ~np.isnan(desc).any(axis=1)          # a view with an empty mask yields NaN and is skipped (:59-65)
    return desc[keep]


def process(model, dataset, todo, args, features_path, rank=0, quiet=False, stamps=None):
    """the per-mesh loop (reference :36-70); returns the number of meshes written.  `stamps` (optional list) receives the host time at
    which each mesh's file was written (bench.py: steady-state interval between meshes)"""
    import time
    done = 0
    for n, idx in enumerate(todo):
        if not quiet:
            print(f"[rank {rank}] processing {idx + 1} / {len(dataset)}", flush=True)
        sample = dataset[idx]
        if n + 1 < len(todo) and not args.no_prefetch:
            dataset.prefetch(todo[n + 1])       # tar reads + PNG decode + host->device copy of the next mesh run under this mesh's ViT calls
        if sample["templates"] is None:
            print(f"skipping {sample['model_name']}", flush=True)
            continue
        desc = mesh_descriptors(model, sample, args.feature, args.layer, args.batch_size)
        np.save((Path(features_path) / f"{sample['model_name']}.npy").as_posix(), desc)
        done += 1
        if stamps is not None:
            stamps.append(time.perf_counter())
    return done


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards_folder", type=str, default="objaverse_shards")
    ap.add_argument("--filelist", type=str, default="mesh_cache.csv")
    ap.add_argument("--feature", type=str, default="ffa", choices=["ffa", "cls"])
    ap.add_argument("--layer", type=int, default=22)
    ap.add_argument("--mesh_per_job", type=int, default=100)
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--n_views", type=int, default=600)                      # not in the reference: views per mesh (600 there)
    ap.add_argument("--model", type=str, default="dinov2_vitl14_reg")        # not in the reference: backbone
    ap.add_argument("--allow_random_weights", action="store_true")           # not in the reference: run without the checkpoint
    ap.add_argument("--gpus", type=int, default=1)                           # not in the reference: self-launch N ranks, one per GPU
    ap.add_argument("--no_prefetch", action="store_true")                    # not in the reference: load every mesh synchronously (A/B, tests)
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    import sys
    parallel.self_launch(args.gpus, ["-m", "scripts.extract_retrieval_features"], sys.argv[1:] if argv is None else list(argv))

    shards_path = Path("data/datasets").resolve() / args.shards_folder
    features_path = Path("data/datasets").resolve() / f"{args.shards_folder}_{args.feature}_{args.layer}"
    features_path.mkdir(parents=True, exist_ok=True)
    filelist_path = Path("data").resolve() / args.filelist

    rank, world, _ = parallel.init_from_env()
    if world > 1:
        parallel.announce("dist")          # backend, RCCL version, device + PCI bus id of every rank
    model = DINOv2FeatureExtractor(args.model, allow_random_weights=args.allow_random_weights or None)
    # (the loop visits each mesh once: the device store of decoded entries only needs the current one and the one on its way)
    dataset = WebTemplateDataset(shards_path.as_posix(), filelist_path.as_posix(), crop=False, n_views=args.n_views, cache_meshes=0)

    if "SLURM_ARRAY_TASK_ID" in os.environ:
        job = int(os.environ["SLURM_ARRAY_TASK_ID"])
        start, end = job * args.mesh_per_job, min((job + 1) * args.mesh_per_job, len(dataset))
        todo = list(range(start, end))[rank::world]
    elif world > 1:
        todo = parallel.shard_items(len(dataset), rank, world)
    else:
        raise KeyError("SLURM_ARRAY_TASK_ID")   # the reference requires it (:32)

    process(model, dataset, todo, args, features_path, rank=rank)
    print("Done", flush=True)


if __name__ == "__main__":
    main()
