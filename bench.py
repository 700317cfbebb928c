"""bench.py — proposals/sec of the per-proposal 6D-pose hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One STEP = one pass of the hot path over `--proposals-per-step` synthetic proposals per GPU, every stage executed:
518x518 crop -> ViT-L/14-reg layer-22 patch features -> FFA -> cosine top-100 over a 46 037 x 1024 bank -> 576 pose
hypotheses of an 82k-triangle mesh rasterised (420^2), cropped, pushed through the ViT, patchwise-scored against the
query -> top-3 -> metric (R,t).  Inputs are resident in HBM before the timed region.  Proposals are sharded across
ranks (weak scaling, no data-path collective); the only collective is the all-gather of 16-float result rows.

Prints ONE JSON line on rank 0 (metric/value/unit/... + "roofline" for the dominant kernel + "cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # HBM3E, same guide


def icosphere(sub: int):
    t = (1 + 5 ** 0.5) / 2
    v = [np.array(p, np.float64) for p in ([-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t],
                                           [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1])]
    v = [p / np.linalg.norm(p) for p in v]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    for _ in range(sub):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (v[a] + v[b]) / 2
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return np.array(v), np.array(f, dtype=np.int32)


def synthetic_mesh(sub: int, seed: int = 40):
    v, f = icosphere(sub)
    v = v * (1 + 0.25 * np.sin(3 * v[:, :1]) * np.cos(2 * v[:, 1:2]) + 0.1 * np.sin(7 * v[:, 2:3]))
    v /= np.abs(v).max()                      # unit half-extent (resize_meshes.py), rendered at scale 0.25
    col = np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=(len(v), 3), dtype=np.uint8)
    return v.astype(np.float32), f, col


def synthetic_bank(N, D, seed=21):
    rng = np.random.Generator(np.random.PCG64(seed))
    mu = rng.standard_normal(D).astype(np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32) + 2.0 * mu      # anisotropic like a real bank
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def synthetic_proposals(B, res, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    crops = torch.from_numpy(rng.random((B, 3, res, res), dtype=np.float32)).to(torch.bfloat16)
    yy, xx = np.mgrid[0:res, 0:res]
    masks = np.stack([(((yy - res * rng.uniform(.4, .6)) / (res * rng.uniform(.2, .45))) ** 2 +
                       ((xx - res * rng.uniform(.4, .6)) / (res * rng.uniform(.2, .45))) ** 2) <= 1 for _ in range(B)])
    boxes = np.array([[100 + 7 * b, 80 + 5 * b, 300 + 9 * b, 260 + 3 * b] for b in range(B)])
    scales = rng.uniform(0.03, 0.15, size=B)
    K = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]], dtype=np.float32)
    return crops, torch.from_numpy(masks), K, boxes, scales


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_time(fn, warmup, iters):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def cpu_baseline(args, mesh_arrays, bank_f32):
    """the CPU restatement of the same workload timed on this box's host cores in the same run — the torch-CPU
    ViT (oracle/vit_ref.py; fp32 and the reference's bf16 regime), the reference's own torch expressions for the bank scan /
    top-100 (scripts/extract_proposals_ground.py:136-140) and the template score (pose_estimator.py:85-90), and the scalar C
    oracle for the rasteriser (the reference renders with OpenGL; no CPU torch path exists).  Bounded sample, medians of
    warmed iterations, extrapolated to one proposal.  A reported baseline, not the optimisation target."""
    import torch.nn.functional as F
    from oracle import fp_oracle as fo, vit_ref
    from freepose_amd.ops import random_state_dict
    sd_bf = random_state_dict("dinov2_vitl14_reg", 0)
    sd = {k: v.float() for k, v in sd_bf.items()}
    x = torch.rand(1, 3, args.res, args.res)
    # pick the thread count that is actually fastest on this box (all logical cores oversubscribes badly on big hosts)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best = (1e30, 1)
    for nt in sorted({min(avail, n) for n in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        dtp = _median_time(lambda: vit_ref.vit_forward(sd, x[:, :, :224, :224], layer=6), 1, 2)
        if dtp < best[0]:
            best = (dtp, nt)
    ncores = best[1]
    # SURVEY 8(d) asks for torch.set_num_threads(os.cpu_count()): reported beside the fastest count (one timed forward)
    all_cores = None
    if avail != ncores:
        torch.set_num_threads(avail)
        t_all = _median_time(lambda: vit_ref.vit_forward(sd_bf, x, layer=22, feature_type="patch", dtype=torch.bfloat16), 0, 1)   # one forward, no warm-up: ~45 s on the pool's hosts
        all_cores = {"cores": avail, "vit_per_crop_bf16": t_all}
    torch.set_num_threads(ncores)
    feats = [None]

    def vit32():
        feats[0] = vit_ref.vit_forward(sd, x, layer=22, feature_type="patch")
    t_vit = _median_time(vit32, 2, 3)
    t_vit_bf16 = _median_time(lambda: vit_ref.vit_forward(sd_bf, x, layer=22, feature_type="patch", dtype=torch.bfloat16), 2, 3)
    # bank scan + top-100: the reference's expressions on the full bank
    rf = F.normalize(torch.from_numpy(bank_f32).to(torch.bfloat16), dim=-1)
    q = F.normalize(torch.randn(1024, generator=torch.Generator().manual_seed(0)).to(torch.bfloat16), dim=-1)
    t_scan = _median_time(lambda: torch.topk((rf @ q).float(), min(100, rf.shape[0])), 2, 5)
    # template score: einsum(normalize(T), normalize(q)).mean(-1) on a slice of the hypotheses
    P = (args.res // 14) ** 2
    Ts = min(32, args.hyp)
    tm = torch.randn(Ts, P, 1024, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)
    qf = feats[0].to(torch.bfloat16)
    t_score = _median_time(lambda: torch.einsum("bnd,bnd->bn", F.normalize(tm, dim=-1), F.normalize(qf, dim=-1)).mean(dim=-1), 1, 3) / Ts * args.hyp
    v, f, c = mesh_arrays
    from freepose_amd.src.pipeline.retrieval.renderer import grid_poses
    poses = np.array(grid_poses(args.hyp))[:2].astype(np.float32)
    t_raster = _median_time(lambda: fo.rasterize(v, f, c, poses, 0.25, 600, 600, 210, 210, 420, 420), 1, 3) / 2 * args.hyp
    per_prop = (1 + args.hyp) * t_vit + t_scan + t_score + t_raster
    per_prop_bf16 = (1 + args.hyp) * t_vit_bf16 + t_scan + t_score + t_raster
    return {"value": 1.0 / per_prop, "unit": "proposals/s", "cores": ncores, "kind": "port",
            "cpu_model": _cpu_model(), "logical_cpus_available": avail, "torch": torch.__version__,
            "sample": f"ViT-L/14 layer-22 forward @{args.res}^2 on {ncores} torch threads: fp32 {t_vit:.2f} s/crop, bf16 {t_vit_bf16:.2f} s/crop "
                      f"(2 warm-ups, median of 3); reference torch expressions for bank scan + top-100 over {bank_f32.shape[0]} rows "
                      f"(median of 5) and template score on {Ts} of {args.hyp} hypotheses (median of 3); 2 renders of the {len(f)}-triangle "
                      f"mesh with the single-thread C oracle (1 warm-up, median of 3); extrapolated to 1+{args.hyp} forwards and "
                      f"{args.hyp} hypotheses; thread count = the fastest of 8/16/32/64/128 on a 6-block 224^2 probe, "
                      f"{avail} logical CPUs available",
            "seconds_per_proposal": per_prop, "value_bf16": 1.0 / per_prop_bf16, "seconds_per_proposal_bf16": per_prop_bf16,
            "all_cores": None if all_cores is None else dict(all_cores, value_bf16=1.0 / ((1 + args.hyp) * all_cores["vit_per_crop_bf16"] + t_scan + t_score + t_raster),
                                                             note="torch.set_num_threads(all logical CPUs): slower than the chosen count on this host; the other stages as above"),
            "stage_seconds": {"vit_per_crop_fp32": t_vit, "vit_per_crop_bf16": t_vit_bf16, "bank_scan_topk": t_scan,
                              "template_score": t_score, "raster": t_raster}}


def video_workload(args, vit, rank, world):
    """BASELINE config 5 as a secondary measurement: `--video-frames` frames of ONE object through
    DinoOnlinePoseEstimator.forward (coarse estimate on the first frame of a stretch, then per frame ~19 neighbour renders,
    crops, ViT, patchwise score).  One rank: the whole clip sequentially (the reference's semantics).  N ranks: contiguous
    frame chunks with a coarse re-initialisation each — SURVEY §8(e) option 4, which DEVIATES from the reference on
    chunk-initial frames and is labelled so."""
    from freepose_amd import ops, parallel
    from freepose_amd.mesh_io import TriMesh
    from freepose_amd.src.pipeline.estimators.online_pose_estimator import DinoOnlinePoseEstimator
    from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer
    from freepose_amd.src.pipeline.utils import Proposals
    import torch.distributed as dist
    n_frames = args.video_frames
    mv, mf, mc = synthetic_mesh(4)                                   # 5 120 triangles, vertex colours
    mesh = TriMesh(mv, mf, mc)
    fe = DINOv2FeatureExtractor.__new__(DINOv2FeatureExtractor)      # share the already-resident ViT-L
    torch.nn.Module.__init__(fe)
    fe.model_name, fe.model, fe.num_register_tokens = "dinov2_vitl14_reg", vit, vit.n_reg
    est = DinoOnlinePoseEstimator(n_coarse_poses=600, n_fine_poses=20000, cache_size=4, cache_dir=f"/tmp/fp_bench_cache_r{rank}",
                                  feature_extractor=fe, hypothesis_cache=0)    # every hypothesis recomputed in every frame: the reference's step
    r600 = MeshRenderer(600)
    renders = r600.render(mesh, scale=0.25)
    crops, _, _ = MeshRenderer.generate_proposals(renders)
    template = {"templates": crops.float(), "depths": renders.depth, "model_name": "bench_mesh",
                "intrinsic": torch.tensor([[600, 0, 210], [0, 600, 210], [0, 0, 1]])}
    # frames: each object drawn at a slowly rotating pose on a noisy 1280x720 background (guessed intrinsics, video :115-118)
    H, W, scale = 720, 1280, 0.10
    f = float(np.sqrt(H ** 2 + W ** 2))
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]])
    from scipy.spatial.transform import Rotation as Rot
    dm = ops.Mesh(mv, mf, mc)
    est.coarse_estimator._get_template_features(template)            # template features resident (the drivers' cache hit path)

    def run_clip(n_obj, frames_total, shard="frames", window=1):
        """`frames_total` frames with `n_obj` tracked objects each; the objects of a frame go through ONE batched step
        (DinoOnlinePoseEstimator.forward_fine_many, what scripts.dino_inference_video does).  shard = "frames": contiguous frame chunks
        per rank (deviating: coarse re-initialisation per chunk); "objects": every rank walks ALL frames with its own objects
        (prev_pose chains stay intact: exact w.r.t. the reference, SURVEY 8e option 1 — what scripts.dino_inference_video does)."""
        mine = parallel.shard_chunk(frames_total, rank, world) if shard == "frames" else list(range(frames_total))
        n_total = n_obj
        my_objs = list(range(n_obj)) if shard == "frames" else parallel.shard_items(n_obj, rank, world)
        meshes = [TriMesh(mv, mf, mc) for _ in range(n_obj)]          # distinct host meshes: one resident device mesh each
        axes = [np.array([0.2, 1.0, 0.1]), np.array([1.0, 0.3, -0.2]), np.array([-0.4, 0.2, 1.0]), np.array([0.6, -0.8, 0.3])]
        gt, props = [], []
        rng = np.random.Generator(np.random.PCG64(3))
        for fr in mine:
            fg, fp = [], []
            for o in my_objs:
                R0 = np.array(est.coarse_estimator.mesh_poses[(37 + 101 * o) % len(est.coarse_estimator.mesh_poses)])[:3, :3]
                ax = axes[o % 4] / np.linalg.norm(axes[o % 4])
                P = np.eye(4)
                P[:3, :3] = Rot.from_rotvec(np.deg2rad(1.5 * fr) * ax).as_matrix() @ R0
                P[:3, 3] = [0.05 + 0.0005 * fr + 0.12 * ((o % 4) - 1.5 * (n_total > 1)), -0.02 + 0.03 * (o % 2), 0.9]
                rgb, depth = ops.rasterize(dm, torch.from_numpy(P[None].astype(np.float32)), scale, f, f, W / 2.0, H / 2.0, W, H)
                m = (depth[0] > 0).cpu().numpy()
                img = rng.integers(0, 50, size=(H, W, 3), dtype=np.uint8)
                img[m] = rgb[0].cpu().numpy()[m]
                ys, xs = np.nonzero(m)
                box = torch.tensor([[int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())]])
                pr = Proposals(img, {"boxes": box, "masks": torch.from_numpy(m[None])}, 420, bbox_extend=0.05)
                fp.append((pr.proposals[0], pr.proposals_masks[0], box[0]))
                fg.append(P)
            props.append(fp)
            gt.append(fg)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        prev, errs = [None] * len(my_objs), []
        qn = {}
        for fi, (fp, fg) in enumerate(zip(props, gt)):
            if not fp:
                continue             # (more ranks than objects: this rank only joins the barriers)
            if window > 1 and fi % window == 0:
                # the query crops of the next `window` frames in ONE ViT call (scripts.dino_inference_video --query_window): a query depends on
                # its frame only; the stretch's first frame goes through the coarse estimator, which encodes its own
                todo = [(g, o) for g in range(fi, min(fi + window, len(props))) for o in range(len(props[g])) if g > 0]
                if todo:
                    crops = torch.stack([torch.as_tensor(props[g][o][0]) for g, o in todo]).to("cuda", torch.bfloat16)
                    feats = ops.l2_normalize(est.feature_extractor(crops, layer=22, feature_type="patch"))
                    qn = {go: feats[i:i + 1] for i, go in enumerate(todo)}
            if prev[0] is None:      # head of the stretch: coarse estimate + fine step per object
                outs = [est(c, cm, template, meshes[my_objs[o]], K, b, scale, prev_pose=None, neighborhood=15, layer=22, batch_size=128)
                        for o, (c, cm, b) in enumerate(fp)]
            else:
                outs = est.forward_fine_many([dict(proposal=c, proposal_mask=cm, template_dict=template, mesh=meshes[my_objs[o]], K=K, bbox=b,
                                                   est_scale=scale, prev_pose=prev[o], query_feat=qn.get((fi, o))) for o, (c, cm, b) in enumerate(fp)],
                                             neighborhood=15, layer=22)
            for o, out in enumerate(outs):
                prev[o] = out["TCO"][0]
                Rr = prev[o][:3, :3] @ fg[o][:3, :3].T
                errs.append(np.degrees(np.arccos(np.clip((np.trace(Rr) - 1) / 2, -1, 1))))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()), len(mine)

    dt1, mine1 = run_clip(1, n_frames)
    n_obj = max(1, args.video_objects)
    frames_multi = max(world, n_frames // n_obj)
    dtm, minem = run_clip(n_obj, frames_multi) if n_obj > 1 else (dt1, mine1)
    # the same clips with the estimator's hypothesis store (its default in the video driver): only the fine-grid hypotheses that ENTER
    # the 15-degree neighbourhood are rendered and sent through the ViT, the others are read from the per-mesh device store — same poses
    # and scores bit for bit (tests/test_gpu_pipeline.py::test_hypothesis_store_gives_the_recomputed_results)
    est.hypothesis_cache = 768
    dt1c, mine1c = run_clip(1, n_frames)
    dtmc, minemc = run_clip(n_obj, frames_multi) if n_obj > 1 else (dt1c, mine1c)
    dt1w, mine1w = run_clip(1, n_frames, window=8)                   # + the query crops of 8 frames in one ViT call (the driver's --query_window 8)
    dtmw, minemw = run_clip(n_obj, frames_multi, window=8) if n_obj > 1 else (dt1w, mine1w)
    est.hypothesis_cache = 0
    strong = None
    if world > 1:                                                     # strong scaling with the chains intact: a fixed set of objects dealt to the ranks
        n_strong, frames_strong = 8, max(10, n_frames // 8)
        dts, _ = run_clip(n_strong, frames_strong, shard="objects")
        strong = {"metric": "frame-objects/s (object-sharded clip: every rank tracks its own objects through ALL frames, prev_pose chains intact)",
                  "value": n_strong * frames_strong / dts, "unit": "frame-objects/s", "objects": n_strong, "frames": frames_strong,
                  "objects_per_rank": [len(parallel.shard_items(n_strong, r, world)) for r in range(world)], "seconds": dts, "scaling": "strong",
                  "note": "exact w.r.t. the reference (SURVEY 8e option 1; scripts.dino_inference_video without --frame_chunks); the total work is fixed, "
                          "ranks with more objects batch them per frame"}
    return {"metric": "frames/sec (dino_inference_video step: 1 object, rescoring)", "value": n_frames / dt1, "unit": "frames/s",
            "frames": n_frames, "ms_per_frame_per_gpu": dt1 / max(mine1, 1) * 1e3,
            "multi_object": {"objects_per_frame": n_obj, "frames": frames_multi, "frame_objects_per_s": frames_multi * n_obj / dtm,
                             "ms_per_frame_object_per_gpu": dtm / max(minem, 1) / n_obj * 1e3,
                             "note": "the objects of a frame share one batched render-and-compare step (one ViT call, one host copy); "
                                     "per-object results equal the one-by-one run bit for bit (tests/test_gpu_cli_e2e.py)"},
            "hypothesis_store": {"frames_per_s": n_frames / dt1c, "ms_per_frame_per_gpu": dt1c / max(mine1c, 1) * 1e3,
                                 "multi_object_ms_per_frame_object_per_gpu": dtmc / max(minemc, 1) / n_obj * 1e3,
                                 "object_rotation_deg_per_frame": 1.5,
                                 "with_query_window_8": {"frames_per_s": n_frames / dt1w, "ms_per_frame_per_gpu": dt1w / max(mine1w, 1) * 1e3,
                                                         "multi_object_ms_per_frame_object_per_gpu": dtmw / max(minemw, 1) / n_obj * 1e3,
                                                         "note": "the driver's default: the query crops of 8 consecutive frames share one ViT call (a query "
                                                                 "crop depends on its frame only); same poses and scores"},
                                 "note": "the same clips with DinoOnlinePoseEstimator(hypothesis_cache=768), the video driver's default: hypotheses already "
                                         "seen for the mesh are read from the device store instead of being re-rendered and re-encoded; identical "
                                         "results.  `value` / `ms_per_frame_per_gpu` above are WITHOUT it (every hypothesis recomputed per frame)"},
            "object_sharded": strong,
            "sharding": "sequential clip on one rank (reference semantics)" if world == 1 else
                        f"{world} contiguous frame chunks, coarse re-initialisation per chunk (SURVEY 8e option 4: DEVIATES from the reference "
                        "on chunk-initial frames)",
            "note": "throughput of the tracking step on seeded random-init weights (no checkpoint offline): poses are not meaningful, "
                    "pose recovery with the same code is asserted in tests/test_gpu_cli_e2e.py and tests/test_gpu_pipeline.py",
            "config": "1280x720 frames, ViT-L/14-reg @420^2, 600 coarse + 20000 fine hypotheses, 15 deg neighbourhood, 5120-triangle mesh"}


def _timed(fn, iters=3, warmup=1):
    """median HIP-event time (ms) of fn() on the current stream"""
    from freepose_amd import ops
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        t = ops.Timer()
        t.start()
        fn()
        t.stop()
        ts.append(t.elapsed_ms())
    return float(np.median(ts))


def config_legs(args, vit, bank):
    """BASELINE configs 2 and 3 as secondary measurements (rank 0, one GPU, outside the timed region).

    config2 — scripts/extract_retrieval_features.py:36-70 (bank building): ViT-L layer-22 patch features -> per-view FFA (fp32 rows)
    -> per-object mean (scripts/merge_features.py:19-35).  Two shapes: B = 256 crops @518^2 in one batch (BASELINE's statement of the
    config: 6 objects x 42 views + 4), and the reference's own loop shape, 600 views @420^2 in batches of 256 / 256 / 88 with one
    device->host copy per mesh.
    config3 — scripts/dino_inference.py:108-111 -> src/pipeline/estimators/pose_estimator.py:55-60,84-92 with the template features
    CACHED (the reference's steady state): the 5 proposals of one image share ONE ViT call and ONE bank pass
    (scripts/extract_proposals_ground.py:136-140), then per proposal: normalised-query patchwise score against the resident
    600 x 900 x 1024 store of its mesh (5 different stores), top-3, depth extents -> pose."""
    from freepose_amd import ops
    from freepose_amd.src.pipeline.estimators.pose_estimator import DinoPoseEstimator
    from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    out = {}
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(10)

    def masks_for(B, res, seed):
        rng = np.random.Generator(np.random.PCG64(seed))
        yy, xx = np.mgrid[0:res, 0:res]
        return torch.from_numpy(np.stack([(((yy - res * rng.uniform(.4, .6)) / (res * rng.uniform(.25, .45))) ** 2 +
                                           ((xx - res * rng.uniform(.4, .6)) / (res * rng.uniform(.25, .45))) ** 2) <= 1 for _ in range(B)])).to(dev)

    # ---------------- config 2 (a): one batch of 256 crops @518^2 -------------------------------------------------------------
    B, res, views = 256, 518, 42
    crops = torch.rand((B, 3, res, res), generator=g).to(torch.bfloat16).to(dev)
    masks = masks_for(B, res, 11)
    gg = res // 14
    st = {}

    def c2a():
        t0, t1 = ops.Timer(), ops.Timer()
        t0.start()
        feats = vit(crops, layer=22, feature_type="patch")
        t0.stop()
        t1.start()
        desc = ops.ffa(feats, masks[:, : gg * 14, : gg * 14], cell=14, out_f32=True)                       # [B, 1024] fp32 per-view rows
        objs = torch.stack([desc[o:o + views].mean(dim=0) for o in range(0, B - views + 1, views)])   # merge_features.py: mean over an object's views
        t1.stop()
        st["vit"], st["ffa"], st["objs"] = t0, t1, objs
        return objs
    ms = _timed(c2a)
    fl = vit.flops(B, res, res, 22)
    out["config2_batch256_518"] = {
        "metric": "crops/s (extract_retrieval_features: ViT-L/14 layer-22 + FFA + per-object mean, one batch of 256 @518^2)",
        "value": B / ms * 1e3, "unit": "crops/s", "ms": ms, "vit_ms": st["vit"].elapsed_ms(), "ffa_mean_ms": st["ffa"].elapsed_ms(),
        "roofline": {"bound": "mfma", "achieved": fl / ms / 1e9, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS},
        "objects": int(st["objs"].shape[0])}
    del crops, masks

    # ---------------- config 2 (b): the reference's loop: 600 views @420^2, batches 256 / 256 / 88, one host copy per mesh ----
    T, res = 600, 420
    tmpl = torch.rand((T, 3, res, res), generator=g).to(torch.bfloat16).to(dev)
    tmask = masks_for(T, res, 12)

    def c2b():
        feats = torch.cat([vit(tmpl[i:i + 256], layer=22, feature_type="patch") for i in range(0, T, 256)], dim=0)
        desc = ops.ffa(feats, tmask, cell=14, out_f32=True).cpu().numpy()                                  # the mesh's .npy rows
        return desc[~np.isnan(desc).any(axis=1)]
    c2b()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        c2b()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / 3
    fl = vit.flops(T, res, res, 22)
    out["config2_mesh600_420"] = {
        "metric": "crops/s (extract_retrieval_features per mesh: 600 views @420^2 in batches 256/256/88, FFA, host copy of the rows)",
        "value": T / sec, "unit": "crops/s", "ms_per_mesh": sec * 1e3, "meshes_per_s": 1.0 / sec,
        "roofline": {"bound": "mfma", "achieved": fl / sec / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS}}

    # ---------------- config 3: cached per-proposal path, the proposals of one image batched ------------------------------------
    n_prop, P, D = 5, 900, 1024
    fe = DINOv2FeatureExtractor.__new__(DINOv2FeatureExtractor)      # share the already-resident ViT-L
    torch.nn.Module.__init__(fe)
    fe.model_name, fe.model, fe.num_register_tokens = "dinov2_vitl14_reg", vit, vit.n_reg
    est = DinoPoseEstimator(n_poses=T, cache_size=n_prop, cache_dir="/tmp/fp_bench_cache_c3", feature_extractor=fe)
    depth = torch.zeros((T, res, res), dtype=torch.float32, device=dev)
    depth[:, 120:300, 140:290] = 1.1                                   # a plausible silhouette per view (extents only need the support)
    tdicts = []
    for m in range(n_prop):
        f = torch.randn((T, P, D), generator=g, dtype=torch.float32).to(torch.bfloat16).to(dev)
        est._cache_features(f"mesh{m}", f)                            # enters the device store normalised in place (1.1 GB each)
        tdicts.append({"model_name": f"mesh{m}", "templates": tmpl, "depths": depth, "intrinsic": torch.tensor([[600, 0, 210], [0, 600, 210], [0, 0, 1]])})
    pcrops, pmasks, K, boxes, scales = synthetic_proposals(n_prop, res, seed=20)
    pcrops, pmasks = pcrops.to(dev), pmasks.to(dev)
    stages = {}

    def span(name):
        t = ops.Timer()
        stages.setdefault(name, []).append(t)
        return t

    def c3():
        stages.clear()
        t = span("vit_query_batch"); t.start()
        feats = vit(pcrops, layer=22, feature_type="patch")                                     # ONE call for the image's proposals
        t.stop()
        t = span("ffa"); t.start()
        desc = ops.ffa(feats, pmasks, cell=14, normalize=True)
        t.stop()
        t = span("bank_scan_topk"); t.start()
        s_, i_ = bank.topk(desc, 100)                                                           # ONE bank pass for all of them
        t.stop()
        t = span("estimator_forward_cached"); t.start()
        res_ = est.forward_many([dict(proposal=pcrops[pidx], template_dict=tdicts[pidx], K=K, bbox=boxes[pidx], est_scale=float(scales[pidx]),
                                      query_feat=feats[pidx:pidx + 1]) for pidx in range(n_prop)])       # what scripts.dino_inference runs per image
        t.stop()
        return res_, s_, i_
    c3()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 5
    for _ in range(iters):
        c3()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / iters
    st_ms = {k: sum(t.elapsed_ms() for t in v) for k, v in stages.items()}

    # the same steady state with the proposals of 8 images per step (scripts.dino_inference --image_window 8): ONE ViT call over the 40 crops,
    # one FFA, 40 queries through the bank, ONE estimator step with one device -> host copy; per-crop results are those of the per-image step
    n_win = 8
    wcrops, wmasks = pcrops.repeat(n_win, 1, 1, 1), pmasks.repeat(n_win, 1, 1)

    def c3_window():
        feats = vit(wcrops, layer=22, feature_type="patch")
        desc = ops.ffa(feats, wmasks, cell=14, normalize=True)
        s_, i_ = bank.topk(desc, 100)
        return est.forward_many([dict(proposal=wcrops[j], template_dict=tdicts[j % n_prop], K=K, bbox=boxes[j % n_prop], est_scale=float(scales[j % n_prop]),
                                      query_feat=feats[j:j + 1]) for j in range(n_win * n_prop)]), s_, i_
    one, _, _ = c3()
    win, _, _ = c3_window()
    same = all(np.array_equal(win[j]["TCO"][0], one[j % n_prop]["TCO"][0]) and float(win[j]["scores"][0]) == float(one[j % n_prop]["scores"][0]) for j in range(n_win * n_prop))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        c3_window()
    torch.cuda.synchronize()
    sec_w = (time.perf_counter() - t0) / 3
    q = ops.l2_normalize(torch.randn((P, D), generator=g).to(torch.bfloat16).to(dev))
    ts_ms = _timed(lambda: ops.template_score(est.feature_cache["mesh0"], q, normalized=True), iters=5)
    vfl = vit.flops(n_prop, res, res, 22)
    passes = -(-n_prop // ops.BANK_QUERIES_PER_PASS) if hasattr(ops, "BANK_QUERIES_PER_PASS") else -(-n_prop // 4)
    out["config3_cached_proposals"] = {
        "metric": "proposals/s (dino_inference steady state: template features cached; 5 proposals of an image share one ViT call and one bank pass)",
        "value": n_prop / sec, "unit": "proposals/s", "ms_per_image": sec * 1e3, "proposals_per_image": n_prop,
        "stages_ms_per_image": st_ms,
        "window_of_8_images": {"proposals_per_s": n_win * n_prop / sec_w, "ms_per_image": sec_w / n_win * 1e3, "proposals_per_step": n_win * n_prop,
                               "same_poses_and_scores_as_per_image": bool(same),
                               "note": "scripts.dino_inference --image_window 8 (its default): the proposals of 8 images share one ViT call, one bank pass "
                                       "set and one estimator step; `value` above is the per-image step"},
        "roofline": {
            "vit_query_batch": {"bound": "mfma", "achieved": vfl / max(st_ms.get("vit_query_batch", 1e9), 1e-9) / 1e9, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": vfl / max(st_ms.get("vit_query_batch", 1e9), 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS},
            "bank_scan_topk": {"bound": "hbm", "achieved": passes * args.bank * D * 2.0 / max(st_ms.get("bank_scan_topk", 1e9), 1e-9) / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": passes * args.bank * D * 2.0 / max(st_ms.get("bank_scan_topk", 1e9), 1e-9) / 1e6 / HBM_PEAK_GBS, "bank_passes": passes},
            "template_score_normed": {"bound": "hbm", "achieved": T * P * D * 2.0 / ts_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": T * P * D * 2.0 / ts_ms / 1e6 / HBM_PEAK_GBS, "ms": ts_ms, "bytes": T * P * D * 2.0}},
        "note": "estimator_forward_cached = DinoPoseEstimator.forward_many over the image's proposals with the caller's query features: per proposal "
                "normalise query, streaming score over the 1.1 GB pre-normalised store, top-3, depth extents of the 3 winners; ONE device -> host copy "
                "per image, then the host pose formula (round 4: one sync per proposal)"}
    est.feature_cache.clear()
    return out


def make_cli_workspace(n_meshes, T=600):
    """a synthetic workspace on disk in the reference's layout (tests/_synth_scene.py): two OBJ meshes rendered by scripts.render_templates
    into `data/datasets/objaverse_shards/shard-000000.tar` (T views each), the members re-used under `n_meshes` names so that every mesh
    is a different set of tar reads and PNG decodes.  Returns (root, mesh names); the caller chdir()s into root and removes it."""
    import io
    import tarfile
    import tempfile
    from tests import _synth_scene as sc
    root = Path(tempfile.mkdtemp(prefix="fp_bench_ws_"))
    cwd = os.getcwd()
    try:
        os.chdir(root)
        sc.write_meshes(root)
        tar_path = sc.render_shards(root, T)
        names = [f"m{i:02d}{sc.MESH_IDS[i % 2]}" for i in range(n_meshes)]
        with tarfile.open(tar_path) as src:
            blobs = {m.name: src.extractfile(m).read() for m in src.getmembers()}
        tar_path.unlink()
        for side in tar_path.parent.glob("*.npy"):
            side.unlink()
        with tarfile.open(tar_path, "w") as dst:
            for i, nm in enumerate(names):
                base = sc.MESH_IDS[i % 2]
                for k in range(T):
                    for suffix in ("rgb.png", "depth.png"):
                        b = blobs[f"{base}_{k}.{suffix}"]
                        info = tarfile.TarInfo(f"{nm}_{k}.{suffix}")
                        info.size = len(b)
                        dst.addfile(info, io.BytesIO(b))
        (root / "data" / "mesh_cache.csv").write_text("model_name\n" + "\n".join(names) + "\n")
    finally:
        os.chdir(cwd)
    return root, names


def cli_legs(args, vit, in_memory_meshes_per_s=None):
    """BASELINE configs 2 and 3 THROUGH THE CLI LOOPS, files on disk (secondary, rank 0): a synthetic workspace in the reference's layout
    (tests/_synth_scene.py: two OBJ meshes -> scripts.render_templates on the HIP rasteriser -> `shard-000000.tar`, 600 views per mesh),
    the shard's members re-used under `--cli-meshes` names so that every mesh is a different set of tar reads and PNG decodes.
      bank build  = scripts.extract_retrieval_features.process(): per mesh tar reads + 1200 PNG decodes + host->device copy + 600 ViT-L
                    forwards @420^2 + FFA + one .npy, with the next mesh's host stage prefetched under this mesh's ViT calls, and the
                    same loop with --no_prefetch;
      inference   = scripts.dino_inference.process_images(): images of two proposals each, every proposal a NEW mesh (cold: template
                    decode + 600 ViT forwards per proposal), then the same images again (warm: template store + feature store hits)."""
    import shutil
    from tests import _synth_scene as sc
    from scripts import dino_inference, extract_retrieval_features
    from freepose_amd.src.dataloader.bop import BOPDataset
    from freepose_amd.src.dataloader.template import WebTemplateDataset
    from freepose_amd.src.pipeline.estimators.pose_estimator import DinoPoseEstimator
    from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    out = {}
    cwd = os.getcwd()
    n_meshes, T = max(2, min(10, args.cli_meshes)), 600
    root, names = make_cli_workspace(n_meshes, T)
    try:
        os.chdir(root)
        fe = DINOv2FeatureExtractor.__new__(DINOv2FeatureExtractor)      # share the already-resident ViT-L
        torch.nn.Module.__init__(fe)
        fe.model_name, fe.model, fe.num_register_tokens = "dinov2_vitl14_reg", vit, vit.n_reg
        shards = root / "data" / "datasets" / "objaverse_shards"
        # ---- bank build -----------------------------------------------------------------------------------------------------
        res = {}
        for mode in ("prefetch", "no_prefetch"):
            a = extract_retrieval_features.build_parser().parse_args(["--batch_size", "256", "--n_views", str(T)] + (["--no_prefetch"] if mode == "no_prefetch" else []))
            fdir = root / "data" / "datasets" / f"feat_{mode}"
            fdir.mkdir(parents=True, exist_ok=True)
            ds = WebTemplateDataset(shards.as_posix(), (root / "data" / "mesh_cache.csv").as_posix(), crop=False, n_views=T, cache_meshes=0)
            extract_retrieval_features.process(fe, ds, [0], a, fdir, quiet=True)          # warm-up: pinned buffers, index, first touches
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stamps = []
            done = extract_retrieval_features.process(fe, ds, list(range(n_meshes)), a, fdir, quiet=True, stamps=stamps)
            torch.cuda.synchronize()
            sec = time.perf_counter() - t0
            steady = float(np.median(np.diff(stamps))) if len(stamps) > 2 else None      # the first mesh's host stage has nothing to hide under
            res[mode] = {"meshes": done, "seconds": sec, "meshes_per_s": done / sec, "host_stage_s_per_mesh": ds.decode_seconds / (done + 1),
                         "steady_state_s_per_mesh": steady, "steady_state_meshes_per_s": None if not steady else 1.0 / steady}
        same = all((root / "data" / "datasets" / "feat_prefetch" / f"{n}.npy").read_bytes() ==
                   (root / "data" / "datasets" / "feat_no_prefetch" / f"{n}.npy").read_bytes() for n in names)
        out["config2_bank_build_cli"] = {
            "metric": "meshes/s (scripts.extract_retrieval_features loop on shards on disk: tar reads + 1200 PNG decodes + H2D + 600 ViT-L forwards @420^2 + FFA + .npy per mesh)",
            "value": res["prefetch"]["meshes_per_s"], "unit": "meshes/s", "meshes": n_meshes, "views_per_mesh": T,
            "prefetch": res["prefetch"], "no_prefetch": res["no_prefetch"], "outputs_byte_identical": bool(same),
            "in_memory_meshes_per_s": in_memory_meshes_per_s,
            "vs_in_memory": None if not in_memory_meshes_per_s else res["prefetch"]["meshes_per_s"] / in_memory_meshes_per_s,
            "steady_state_vs_in_memory": None if not (in_memory_meshes_per_s and res["prefetch"]["steady_state_meshes_per_s"]) else res["prefetch"]["steady_state_meshes_per_s"] / in_memory_meshes_per_s,
            "host_threads": os.cpu_count()}
        # ---- inference: cold and warm ---------------------------------------------------------------------------------------
        n_img = n_meshes // 2
        frames, props, gts, K = sc.draw_frames(root, n_img, T)
        for fr in range(n_img):                                           # every proposal of every image names a mesh of its own
            for o, e in enumerate(props[fr]):
                e["mesh"] = names[2 * fr + o]
        sc.write_bop(root, "synth", frames, props, K)
        flat = json.loads((root / "data" / "results" / "synth" / "props.json").read_text())
        a = dino_inference.build_parser().parse_args(["--dataset", "synth", "--proposals", "props.json", "--n_views", str(T)])
        dataset = BOPDataset("data/datasets/synth/", "test")
        templates = WebTemplateDataset(shards.as_posix(), "data/mesh_cache.csv", bbox_extend=a.bbox_extend, n_views=T, cache_meshes=n_meshes)
        model = DinoPoseEstimator(n_poses=T, cache_size=n_meshes, cache_dir=root / "cache", feature_extractor=fe)
        legs = {}
        for name in ("cold", "warm"):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            # warm: the scene's images eight times over (stores resident) — long enough for the loop's read-ahead and 8-image windows to matter
            rows = dino_inference.process_images(model, templates, dataset, flat, list(range(len(dataset))) * (8 if name == "warm" else 1), a)
            torch.cuda.synchronize()
            sec = time.perf_counter() - t0
            legs[name] = {"proposals": len(rows), "seconds": sec, "proposals_per_s": len(rows) / sec}
        out["config3_dino_inference_cli"] = {
            "metric": "proposals/s (scripts.dino_inference loop on a BOP-layout scene + shards on disk; cold = every proposal a new mesh: template decode + 600 ViT-L forwards; warm = template and feature stores resident)",
            "value": legs["warm"]["proposals_per_s"], "unit": "proposals/s", "cold": legs["cold"], "warm": legs["warm"], "images": n_img, "warm_passes": 8, "proposals_per_image": 2}
        model.feature_cache.clear()
    finally:
        os.chdir(cwd)
        shutil.rmtree(root, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--proposals-per-step", type=int, default=1, help="proposals per GPU per step")
    ap.add_argument("--hyp", type=int, default=576)
    ap.add_argument("--res", type=int, default=518)
    ap.add_argument("--bank", type=int, default=46037)
    ap.add_argument("--mesh-sub", type=int, default=6, help="icosphere subdivisions (6 -> 81 920 triangles)")
    ap.add_argument("--vit-batch", type=int, default=288)   # largest batch whose fc1 output stays below the 4 GiB tile-offset limit; 192 -> 288: -0.6 % (profiles/r04_ab.md)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--video-frames", type=int, default=300,
                    help="frames of the secondary video-tracking measurement (BASELINE config 5; 0 = skip)")
    ap.add_argument("--video-objects", type=int, default=4,
                    help="tracked objects per frame in the multi-object leg of the video measurement (batched per frame)")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the secondary BASELINE config 2 / 3 measurements")
    ap.add_argument("--no-cli-legs", action="store_true", help="skip the config 2 / 3 measurements that go through the CLI loops with files on disk")
    ap.add_argument("--cli-meshes", type=int, default=6, help="meshes of the CLI legs' synthetic shard (2 .. 10)")
    ap.add_argument("--lab", action="store_true",
                    help="tools/ only: load libfreepose_hip_lab.so (measurement variants, FP_* toggles); never a reported number")
    ap.add_argument("--ln-fused", type=int, default=-1, help="fp_ctx_set_option ln_fused (A/B of the LayerNorm fold)")
    args = ap.parse_args()

    if args.lab:
        from freepose_amd import _lib
        _lib.use_lab()
    from freepose_amd import ops, parallel
    from freepose_amd.pipeline import HotPath, StageClock, pack_results
    from freepose_amd.retrieval import TemplateBank
    import torch.distributed as dist

    # plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU) and exit with their status
    parallel.self_launch(args.gpus, [str(Path(__file__).resolve())], sys.argv[1:])
    # stdout carries EXACTLY one line, the JSON result of rank 0: whatever the legs print on the way (the CLI mains report their
    # progress with print()) goes to stderr
    result_stream, sys.stdout = sys.stdout, sys.stderr
    # FP_DIST_BACKEND=gloo lets several ranks share one GPU (flow test on a single-GPU box); the default is RCCL ("nccl")
    rank, world, local = parallel.init_from_env(os.environ.get("FP_DIST_BACKEND", "nccl"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local % torch.cuda.device_count())

    if args.ln_fused >= 0:
        ops.set_option("ln_fused", args.ln_fused)
    vit = ops.ViT("dinov2_vitl14_reg", seed=0)                         # random-init weights of the real architecture
    bank_f32 = synthetic_bank(args.bank, 1024)
    bank = TemplateBank(bank_f32, shard=False)                         # bank replicated, proposals sharded
    mv, mf, mc = synthetic_mesh(args.mesh_sub)
    hp = HotPath(vit, bank, ops.Mesh(mv, mf, mc), n_hyp=args.hyp, crop_res=args.res, vit_batch=args.vit_batch)
    B = args.proposals_per_step
    crops, masks, K, boxes, scales = synthetic_proposals(B, args.res, seed=100 + rank)
    crops, masks = crops.cuda(), masks.cuda()

    def step():
        res = hp.run(crops, masks, K, boxes, scales)
        rows = pack_results(res).cuda()
        return parallel.all_gather_rows(rows) if world > 1 else rows

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    vit.profile(True)
    hp.clock = StageClock()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = vit.profile_read()
    vit.profile(False)
    stage_ms = hp.clock.read()
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dt_ranks = [dt]
    if world > 1:
        parts = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)                                     # every rank's own clock: a straggler shows in the one line
        dt_ranks = [float(x.item()) for x in parts]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    who = parallel.rank_report()                                       # backend, RCCL version, device + PCI bus id of every rank
    video = video_workload(args, vit, rank, world) if args.video_frames > 0 else None    # outside the timed region
    legs = config_legs(args, vit, bank) if (world == 1 and not args.no_config_legs) else None   # BASELINE configs 2 and 3 (secondary)
    if world == 1 and not args.no_cli_legs:                            # the same two configs through the CLI loops, files on disk
        legs = dict(legs or {})
        legs.update(cli_legs(args, vit, (legs.get("config2_mesh600_420") or {}).get("meshes_per_s")))

    if rank == 0:
        n_prop = world * B * args.steps
        flops_vit = vit.flops(1, args.res, args.res, 22) * (1 + args.hyp) * B * args.steps
        gemm_tf = prof["gemm_flops"] / max(prof["ms_gemm"], 1e-9) / 1e9
        out = {
            "metric": "proposals/sec (ViT-L feat + top-k + 576-pose render-compare)",
            "value": n_prop / dt, "unit": "proposals/s",
            # one process per GPU: n_gpus is the number of DISTINCT devices the ranks run on (== world unless FP_ALLOW_SHARED_GPU=1
            # let ranks share a device on a test box; such a line says shared_devices = true and is not a scaling measurement)
            "n_gpus": who["devices_distinct"], "n_ranks": world, "shared_devices": who["shared_devices"], "backend": who["backend"],
            "devices_distinct": who["devices_distinct"], "rccl_version": who["rccl_version"], "comm_stack": who.get("comm_stack", "torch"), "ranks": who["ranks"],
            "ms_per_step_ranks": {"min": min(dt_ranks) / args.steps * 1e3, "median": float(np.median(dt_ranks)) / args.steps * 1e3,
                                  "max": max(dt_ranks) / args.steps * 1e3, "all": [x / args.steps * 1e3 for x in dt_ranks]},
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "library": "libfreepose_hip_lab.so — LAB BUILD, not a reportable number" if args.lab else "libfreepose_hip.so",
            "config": {"workload": f"per-proposal hot path: ViT-L/14-reg layer-22 @{args.res}^2 -> FFA -> top-100 over "
                                   f"{args.bank}x1024 bank -> {args.hyp} hypotheses rasterised ({len(mf)} triangles, 420^2), cropped to "
                                   f"{args.res}^2, ViT + patchwise score -> top-3 pose; all stages per proposal (no feature cache)",
                       "proposals_per_step_per_gpu": B, "vit_forwards_per_proposal": 1 + args.hyp, "weights": "seeded random init, DINOv2 ViT-L/14-reg shapes",
                       "parallelism": f"proposals sharded over {world} rank(s), bank replicated"},
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_kernel + gemm_asm_kernel (all ViT linear layers)", "achieved": gemm_tf,
                         "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tf / MFMA_BF16_PEAK_TFLOPS,
                         "traffic": _pmc_traffic()[0], "traffic_source": _pmc_traffic()[1], "csrc_sha16": csrc_hash(), "launches": prof["gemm_launches"],
                         "avg_launch_ms": prof["ms_gemm"] / max(prof["gemm_launches"], 1),
                         "flops_per_launch": prof["gemm_flops"] / max(prof["gemm_launches"], 1),
                         # the second MFMA-bound kernel of the step (attn_fwd_kernel, ~19 % of it): same definition — algorithmic
                         # 4 n^2 D flops per block and crop over its HIP-event time on the launch stream
                         "attention": _attention_roofline(args, prof, B * args.steps)},
            "stage_ms_rank0": {"vit_gemm": prof["ms_gemm"] / args.steps, "vit_attention": prof["ms_attn"] / args.steps,
                               "vit_other": prof["ms_other"] / args.steps},
            "vit_tflops_end_to_end": flops_vit / dt / 1e12,
            "stages_rank0": stage_table(args, prof, stage_ms, n_tri=len(mf), n_vert=len(mv), n_prop=B * args.steps),
        }
        out["video_workload"] = video
        out["configs"] = legs
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, (mv, mf, mc), bank_f32)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), file=result_stream, flush=True)
    sys.stdout = result_stream
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _attention_roofline(args, prof, n_prop):
    g = args.res // 14
    n_tok, crops = g * g + 5, (1 + args.hyp) * n_prop
    fl = 22 * 4.0 * n_tok * n_tok * 1024 * crops
    ach = fl / max(prof["ms_attn"], 1e-9) / 1e9
    return {"bound": "mfma", "kernel": "attn_fwd_kernel (softmax(QK^T/8)V, 16 heads x 64)", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS, "ms": prof["ms_attn"], "flops": fl}


def stage_table(args, prof, stage_ms, n_tri, n_vert, n_prop):
    """per-stage time (HIP events, rank 0, summed over the timed steps) with the ALGORITHMIC work of SURVEY.md §8(d) and
    the fraction of the roofline that bounds the stage"""
    g = args.res // 14
    P, n_tok, D, H = g * g, g * g + 5, 1024, args.hyp
    crops = (1 + H) * n_prop
    rows = []

    def add(name, ms, bound, work, note=""):
        if ms <= 0:
            return
        if bound == "mfma":
            ach, peak, unit = work / ms / 1e9, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = work / ms / 1e6, HBM_PEAK_GBS, "GB/s"
        rows.append({"stage": name, "ms": ms, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                     "frac": ach / peak, "work": work, "note": note})

    add("vit_linear_layers", prof["ms_gemm"], "mfma", prof["gemm_flops"], "22 x (qk, v, proj, fc1, fc2) + patch embed")
    add("vit_attention", prof["ms_attn"], "mfma", 22 * 4.0 * n_tok * n_tok * D * crops, "4 n^2 D flops per block per crop")
    ln_fused = args.ln_fused != 0   # the product default is on (fp_ctx_set_option "ln_fused"); the library reads no environment variable
    if ln_fused:   # LN1 / LN2 live in the GEMMs: what is left is the block-0 statistics pass, 43 finalisations of 16 partials, final norm, im2col
        other_bytes = (n_tok * D * 2.0 + 43 * n_tok * (128.0 + 20.0) + 2.0 * P * D * 2 + 3.0 * args.res * args.res * 2 + P * 640 * 2.0) * crops
        note = "LayerNorm 1/2 are folded into the GEMMs; this stage = block-0 row statistics, statistics finalisation, final norm + slice, im2col, token init"
    else:
        other_bytes = (22 * 2 + 1) * 2.0 * n_tok * D * 2 * crops + 3.0 * 518 * 518 * 2 * crops
        note = "LayerNorm read+write per call, im2col, token init"
    add("vit_layernorm_etc", prof["ms_other"], "hbm", other_bytes, note)
    add("ffa", stage_ms.get("ffa", 0), "hbm", P * D * 2.0 * n_prop,
        "P*D*2 bytes per crop; cell-mask pooling + blocked masked mean (32 patch blocks x 16 column slabs per crop) + normalise: three small launches, latency-bound at one crop")
    add("bank_scan_topk", stage_ms.get("bank_scan_topk", 0), "hbm", args.bank * D * 2.0 * args.steps,
        f"one pass over the bf16 bank per step, shared by its Q = {args.proposals_per_step} queries; the stage is scan + exact "
        "top-100 select + merge (single-query select ~17 us, latency-bound); the scan kernel alone, measured with the bank evicted / "
        "resident: profiles/r03_scan_cold_warm.log (18.2 us = 5.2 TB/s after a clean 1 GiB read sweep, 15.7 us = 6.0 TB/s resident in the "
        "Infinity Cache; in this pipeline, behind the ViT's dirty activations: see the bank_scan_kernel row of profiles/r04_bench_kernel_stats.csv)")
    add("rasterize", stage_ms.get("rasterize", 0), "hbm", (H * 420 * 420 * 3.0 + n_vert * 32.0 + n_tri * 12.0) * n_prop,
        f"mandatory rgb writes (the depth image is not written: boxes + cloud extents come from the tile epilogue, fp_rasterize_extents); the kernels "
        f"are bound by vector instructions (triangle set-up, coverage, depth test, shading), not by these bytes: profiles/r06_ab.md §1; "
        f"{H * n_tri * n_prop / max(stage_ms.get('rasterize', 1e9), 1e-9) / 1e6:.1f} G triangle set-ups/s")
    add("depth_extents", stage_ms.get("depth_extents", 0), "hbm", H * 420 * 420 * 4.0 * n_prop, "depth read")
    add("crop_resize", stage_ms.get("crop_resize", 0), "hbm", H * 3.0 * args.res * args.res * 2 * n_prop, "bf16 crop writes (reads are a subset of the renders)")
    add("template_score", stage_ms.get("template_score", 0), "hbm", H * P * D * 2.0 * n_prop, "T*P*D*2 bytes")
    add("hypothesis_top3", stage_ms.get("hypothesis_top3", 0), "hbm", H * 8.0 * n_prop, "latency-bound (576 scores)")
    return rows


def csrc_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources: ties a committed rocprofv3 summary to the code it measured
    (the GPU box has no .git, so the commit id itself is not available to bench.py there)"""
    import hashlib
    h = hashlib.sha256()
    for p in sorted((ROOT / "freepose_amd" / "csrc").glob("*")):
        if p.suffix in (".hip", ".h"):
            h.update(p.name.encode())
            h.update(p.read_bytes())
    return h.hexdigest()[:16]


def _pmc_traffic():
    """(HBM bytes per GEMM launch, where the figure comes from): a QUOTED constant — the committed rocprofv3 --pmc summary
    (tools/profile_job.sh -> profiles/), used only if it was taken on exactly these kernel sources (its csrc_sha16 equals
    csrc_hash()); PMC counters cannot be collected inside this run.  (None, reason) otherwise."""
    for name in ("r06_gemm_pmc.json", "r05_gemm_pmc.json", "r04_gemm_pmc.json", "r03_gemm_pmc.json", "r02_gemm_pmc.json"):
        p = ROOT / "profiles" / name
        if p.exists():
            try:
                d = json.loads(p.read_text())
                if d.get("csrc_sha16") == csrc_hash():
                    return d.get("hbm_bytes_per_launch"), f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE passes of tools/profile_job.sh on these sources, csrc_sha16 {csrc_hash()}); quoted, not counted in this run"
            except Exception:
                return None, f"profiles/{name} unreadable"
    return None, "no committed PMC summary matches these kernel sources (csrc_sha16 " + csrc_hash() + ")"


if __name__ == "__main__":
    main()
