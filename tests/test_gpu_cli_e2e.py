"""End-to-end execution of the drop-in CLIs (BASELINE configs 2, 3, 5) on a synthetic workspace in the reference's
on-disk layout: scripts.render_templates -> scripts.extract_retrieval_features -> scripts.merge_features -> TemplateBank, and
proposals JSON -> scripts.dino_inference / scripts.dino_inference_video -> pose CSV, with 1 rank and with 2 ranks sharing the
GPU (gloo): identical CSVs, reference row order and format, poses close to the poses the frames were drawn from."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from tests import _synth_scene as sc

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
N_VIEWS, N_FRAMES, MODEL = 64, 6, "dinov2_vits14_reg"
COLS = ["scene_id", "im_id", "obj_id", "score", "R", "t", "bbox_visib", "scale", "time"]


@pytest.fixture(scope="module")
def workspace(tmp_path_factory):
    root = tmp_path_factory.mktemp("ws")
    sc.write_meshes(root)
    tar = sc.render_shards(root, N_VIEWS)
    assert tar.exists()
    frames, props, gts, K = sc.draw_frames(root, N_FRAMES, N_VIEWS)
    sc.write_video(root, "clip", frames, props)
    sc.write_bop(root, "synth", frames[:2], props[:2], K, depths=sc.draw_frames.depths[:2])
    return root, gts, K


def _run_ranks(module, argv, cwd, world, port):
    env = dict(os.environ, FP_DIST_BACKEND="gloo", FP_ALLOW_SHARED_GPU="1", MASTER_ADDR="127.0.0.1", PYTHONPATH=str(ROOT), SLURM_ARRAY_TASK_ID="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", module] + argv
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r


def _pose(row):
    R = np.array([float(x) for x in row["R"].split()]).reshape(3, 3)
    t = np.array([float(x) for x in row["t"].split()])
    return R, t


def test_video_driver_one_rank_two_ranks_and_modes(workspace, monkeypatch):
    root, gts, K = workspace
    monkeypatch.chdir(root)
    from scripts import dino_inference_video as div
    argv = ["--video", "clip", "--proposals", "props.json", "--n_views", str(N_VIEWS), "--model", MODEL, "--n_fine_poses", "20000",
            "--bbox_extend", "0.05", "--allow_random_weights"]
    div.run(argv)
    out = root / "data" / "results" / "videos" / "clip" / "props_dinopose_layer_22_bbext_0.05_depth_zoedepth.csv"
    df = pd.read_csv(out)
    # reference schema and order: frame-major, objects in proposal order, scene 0, t in metres, time -1 (video :160-176)
    assert list(df.columns) == COLS and len(df) == N_FRAMES * 2
    assert df["im_id"].tolist() == [f for f in range(N_FRAMES) for _ in range(2)]
    assert df["obj_id"].tolist() == sc.MESH_IDS * N_FRAMES and (df["scene_id"] == 0).all() and (df["time"] == -1).all()
    assert df["scale"].tolist() == [0.10, 0.08] * N_FRAMES
    errs = []
    for i, row in df.iterrows():
        R, t = _pose(row)
        fr, o = int(row["im_id"]), i % 2
        # scores are mean patch cosines, except on frame 0 where the reference scores against the UN-normalised query features
        # (online_pose_estimator.py:40-41,76) — reproduced, so only positivity is universal
        assert np.isfinite(R).all() and abs(np.linalg.det(R) - 1) < 1e-6 and np.isfinite(float(row["score"])) and float(row["score"]) > 0
        assert fr == 0 or float(row["score"]) <= 1.0
        errs.append((sc.rotation_error_deg(R, gts[fr, o][:3, :3]), float(np.linalg.norm(t - gts[fr, o][:3, 3]))))
        x, y, w, h = [int(v) for v in row["bbox_visib"].split()]
        assert w > 20 and h > 20 and 0 <= x < 640 and 0 <= y < 480
    errs = np.array(errs)
    # the frames were drawn by the same renderer from grid poses: render-and-compare must land near them (coarse grid of 64
    # views is ~55 deg apart, the fine stage works inside 15 deg of the previous pose)
    assert np.median(errs[:, 0]) < 25.0 and np.median(errs[:, 1]) < 0.15, errs
    text_1 = out.read_text()

    # ---- the run above decoded the next two frames on a background thread; the sequential loop writes the same file -------------------
    out.unlink()
    div.run(argv + ["--read_ahead", "0"])
    assert out.read_text() == text_1
    # ... and kept the fine-grid hypotheses of each mesh between frames; recomputing all of them in every frame (the reference) too
    out.unlink()
    div.run(argv + ["--hypothesis_cache", "0"])
    assert out.read_text() == text_1
    # ... and encoded the query crops of 8 frames per ViT call; one call per frame (and everything off at once) writes the same file
    out.unlink()
    div.run(argv + ["--query_window", "1"])
    assert out.read_text() == text_1
    out.unlink()
    div.run(argv + ["--query_window", "3", "--read_ahead", "0", "--hypothesis_cache", "0"])
    assert out.read_text() == text_1

    # ---- two ranks on the one GPU: objects are sharded, rows all-gathered -> byte-identical CSV -------------------------------
    out.unlink()
    _run_ranks("scripts.dino_inference_video", argv, root, 2, 29571)
    assert out.read_text() == text_1

    # ---- --no_rescore: coarse estimator per frame, frames sharded across ranks; 1 rank == 2 ranks ----------------------------
    out.unlink()
    div.run(argv + ["--no_rescore"])
    text_c1 = out.read_text()
    out.unlink()
    div.run(argv + ["--no_rescore", "--query_window", "1", "--read_ahead", "0"])      # per-frame query forwards: the same file
    assert out.read_text() == text_c1
    out.unlink()
    _run_ranks("scripts.dino_inference_video", argv + ["--no_rescore"], root, 2, 29572)
    assert out.read_text() == text_c1
    dfc = pd.read_csv(out)
    assert len(dfc) == N_FRAMES * 2 and dfc["im_id"].tolist() == df["im_id"].tolist()
    # frame 0 of the rescoring run starts from exactly this coarse estimate (its fine stage then moves inside 15 degrees)
    for o in range(2):
        Rc, _ = _pose(dfc.iloc[o])
        Rf, _ = _pose(df.iloc[o])
        assert sc.rotation_error_deg(Rc, Rf) < 15.0 + 1e-6

    # ---- --frame_chunks (deviating mode): chunk 0 equals the sequential run, later chunks re-initialise coarsely ------------
    _run_ranks("scripts.dino_inference_video", argv + ["--frame_chunks"], root, 2, 29573)
    outk = out.with_name(out.name.replace(".csv", "_chunked2.csv"))
    dfk = pd.read_csv(outk)
    assert len(dfk) == N_FRAMES * 2 and dfk["im_id"].tolist() == df["im_id"].tolist()
    half = (N_FRAMES // 2) * 2
    assert dfk.iloc[:half].to_csv(index=False) == df.iloc[:half].to_csv(index=False)


def test_image_driver_and_bank_build(workspace, monkeypatch):
    root, gts, K = workspace
    monkeypatch.chdir(root)
    monkeypatch.setenv("SLURM_ARRAY_TASK_ID", "0")
    from scripts import dino_inference, extract_retrieval_features, merge_features
    # ---- BASELINE config 3: dino_inference on a BOP-layout scene ---------------------------------------------------------
    argv = ["--dataset", "synth", "--proposals", "props.json", "--n_views", str(N_VIEWS), "--model", MODEL, "--bbox_extend", "0.05",
            "--allow_random_weights"]
    out = dino_inference.run(argv)
    assert out.name == "pose_outputs_0.csv" and "props_dinopose_layer_22_bbext_0.05_depth_zoedepth_cache_50" in str(out)
    df = pd.read_csv(out)
    assert list(df.columns) == COLS and len(df) == 4 and (df["scene_id"] == 48).all() and df["im_id"].tolist() == [1, 1, 2, 2]
    assert (df["time"] == 0.2).all() and df["obj_id"].tolist() == sc.MESH_IDS * 2
    for i, row in df.iterrows():
        R, t_mm = _pose(row)
        fr, o = int(row["im_id"]) - 1, i % 2
        assert abs(np.linalg.det(R) - 1) < 1e-6
        # t is written in millimetres (dino_inference.py:124); z within 20 % of the drawn pose
        assert abs(t_mm[2] / 1000.0 - gts[fr, o][2, 3]) < 0.2 * gts[fr, o][2, 3]
    text_1 = out.read_text()
    assert dino_inference.run(argv + ["--read_ahead", "0"]).read_text() == text_1        # frames read ahead on a thread or not: the same file
    # the run above sent the proposals of up to eight images through one ViT call and one estimator step; per image / odd windows too
    assert dino_inference.run(argv + ["--image_window", "1"]).read_text() == text_1
    assert dino_inference.run(argv + ["--image_window", "3", "--read_ahead", "0"]).read_text() == text_1
    # --depth_method depthmap (reference :82-85): the scale column is the depth-map estimate under each proposal mask — about the
    # objects' drawn scales (0.10 m ball, 0.08 m cube: half the largest extent of the eroded visible surface)
    out_d = dino_inference.run(argv + ["--depth_method", "depthmap"])
    assert "depth_depthmap" in str(out_d)
    dfd = pd.read_csv(out_d)
    from freepose_amd.src.dataloader.bop import BOPDataset
    from freepose_amd.src.pipeline.estimators.scale_estimators import depthmap_scale
    from freepose_amd.src.pipeline.utils import rle_to_mask
    props = json.loads((root / "data" / "results" / "synth" / "props.json").read_text())
    ds = BOPDataset(str(root / "data" / "datasets" / "synth"), "test")
    for i, row in dfd.iterrows():
        entry = ds[int(row["im_id"]) - 1]
        pr = [p for p in props if p["image_id"] == int(row["im_id"])][i % 2]
        assert np.isclose(row["scale"], depthmap_scale(entry["depth"], entry["intrinsic"], rle_to_mask(pr["segmentation"])), rtol=1e-12, atol=0)
        assert 0.5 * pr["scale"] < row["scale"] < 1.3 * pr["scale"], (row["scale"], pr["scale"])
        assert abs(np.linalg.det(_pose(row)[0]) - 1) < 1e-6
    # two ranks: images are dealt round-robin, one CSV per rank; together they hold the same rows
    # the CLI's own launcher: `python -m scripts.dino_inference ... --gpus 2` with no torch.distributed.run around it
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PYTHONPATH=str(ROOT), SLURM_ARRAY_TASK_ID="0", FP_ALLOW_SHARED_GPU="1")
    r = subprocess.run([sys.executable, "-m", "scripts.dino_inference", *argv, "--gpus", "2"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    parts = [pd.read_csv(out.with_name(f"pose_outputs_0_r{r}.csv")) for r in range(2)]
    merged = pd.concat(parts).sort_values(["im_id"], kind="stable").reset_index(drop=True)
    assert merged.to_csv(index=False) == pd.read_csv(out).to_csv(index=False) and text_1

    # ---- BASELINE config 2: extract_retrieval_features (FFA, per-view descriptors) + merge_features -> bank ----------------
    extract_retrieval_features.main(["--filelist", "mesh_cache.csv", "--feature", "ffa", "--layer", "22", "--batch_size", "32",
                                     "--n_views", str(N_VIEWS), "--model", MODEL, "--allow_random_weights"])
    fdir = root / "data" / "datasets" / "objaverse_shards_ffa_22"
    per_mesh = {m: np.load(fdir / f"{m}.npy") for m in sc.MESH_IDS}
    for m, d in per_mesh.items():
        assert d.dtype == np.float32 and d.shape == (N_VIEWS, 384) and np.isfinite(d).all()
    # the loop above prefetched mesh 2 under mesh 1's ViT calls; the sequential loop writes the same files byte for byte
    first = {m: (fdir / f"{m}.npy").read_bytes() for m in sc.MESH_IDS}
    extract_retrieval_features.main(["--filelist", "mesh_cache.csv", "--feature", "ffa", "--layer", "22", "--batch_size", "32",
                                     "--n_views", str(N_VIEWS), "--model", MODEL, "--allow_random_weights", "--no_prefetch"])
    assert all((fdir / f"{m}.npy").read_bytes() == first[m] for m in sc.MESH_IDS)
    merge_features.main(["--features_folder", "objaverse_shards_ffa_22", "--filelist", "mesh_cache.txt"])
    bank = np.load(root / "data" / "objaverse_shards_ffa_22.npy")
    assert bank.shape == (2, 384) and bank.dtype == np.float32
    assert np.allclose(bank[0], per_mesh["balla"].mean(axis=0)) and np.allclose(bank[1], per_mesh["cubet"].mean(axis=0))
    from freepose_amd.retrieval import TemplateBank
    tb = TemplateBank.from_files(root / "data" / "objaverse_shards_ffa_22.npy", root / "data" / "objaverse_shards_ffa_22.ids.txt")
    assert tb.N == 2 and tb.mesh_ids == ["balla", "cubet"]


def test_retrieve_meshes_feeds_the_pose_drivers(workspace, monkeypatch):
    """the retrieval half of the reference's proposal drivers (extract_proposals_ground.py:118-163, ..._video.py:118-197) as a CLI:
    detections JSON (boxes + RLE masks, no mesh) -> ViT -> FFA -> bank top-100 [-> per-view re-rank] [-> soft vote over the frames] ->
    proposals JSON with `mesh` / `score`, which scripts.dino_inference[_video] then consume"""
    root, gts, K = workspace
    monkeypatch.chdir(root)
    monkeypatch.setenv("SLURM_ARRAY_TASK_ID", "0")
    from scripts import dino_inference, dino_inference_video, extract_retrieval_features, merge_features, retrieve_meshes
    if not (root / "data" / "objaverse_shards_ffa_22.npy").exists():         # (the bank of the previous test, when run alone)
        extract_retrieval_features.main(["--filelist", "mesh_cache.csv", "--feature", "ffa", "--layer", "22", "--batch_size", "32",
                                         "--n_views", str(N_VIEWS), "--model", MODEL, "--allow_random_weights"])
        merge_features.main(["--features_folder", "objaverse_shards_ffa_22", "--filelist", "mesh_cache.txt"])
    common = ["--retrieval", "objaverse_shards_ffa_22", "--filelist", "mesh_cache.txt", "--model", MODEL, "--allow_random_weights"]
    # ---- static images ------------------------------------------------------------------------------------------------------
    rd = root / "data" / "results" / "synth"
    props = json.loads((rd / "props.json").read_text())
    dets = [{k: v for k, v in p.items() if k not in ("mesh", "score")} for p in props]
    (rd / "dets.json").write_text(json.dumps(dets))
    out0 = retrieve_meshes.run(["--dataset", "synth", "--detections", "dets.json"] + common)
    assert out0.name == "props-ground-box-0.3-text-0.5-ffa-22-top-0_synth-test.json"
    got0 = json.loads(out0.read_text())
    out5 = retrieve_meshes.run(["--dataset", "synth", "--detections", "dets.json", "--topk", "5", "--output", "props_top5.json"] + common)
    got5 = json.loads(out5.read_text())
    # (the crops of up to 8 images went through one ViT call; image by image gives the same file)
    monkeypatch.setattr(retrieve_meshes, "WINDOW", 1)
    assert retrieve_meshes.run(["--dataset", "synth", "--detections", "dets.json", "--output", "props_w1.json"] + common).read_text() == out0.read_text()
    monkeypatch.setattr(retrieve_meshes, "WINDOW", 8)
    for got in (got0, got5):
        assert len(got) == len(props)
        for g_, p in zip(got, props):
            assert g_["bbox"] == p["bbox"] and g_["segmentation"] == p["segmentation"] and g_["scene_id"] == p["scene_id"] and g_["image_id"] == p["image_id"]
            assert g_["mesh"] in sc.MESH_IDS and np.isfinite(g_["score"]) and g_["time"] == 0.01
    # the coarse pick is the bank row with the best cosine: reproduce it from the files with the product's own bank object
    from freepose_amd.retrieval import TemplateBank
    tb = TemplateBank.from_files(root / "data" / "objaverse_shards_ffa_22.npy", root / "data" / "mesh_cache.txt")
    assert tb.N == 2
    # chained into the pose driver (the scale field comes from compute_scale.py in the reference: a constant here)
    csv = dino_inference.run(["--dataset", "synth", "--proposals", out0.name, "--n_views", str(N_VIEWS), "--model", MODEL, "--bbox_extend", "0.05",
                              "--allow_random_weights", "--depth_method", "const-0.1"])
    df = pd.read_csv(csv)
    assert list(df.columns) == COLS and len(df) == len(props) and df["obj_id"].tolist() == [g_["mesh"] for g_ in got0]
    # ---- video: soft vote over the frames, every frame carries the clip's meshes ------------------------------------------------
    vd = root / "data" / "results" / "videos" / "clip"
    vprops = json.loads((vd / "props.json").read_text())
    (vd / "dets.json").write_text(json.dumps([{k: v for k, v in p.items() if k not in ("mesh", "score")} for p in vprops]))
    outs = {}
    for topk in (0, 5):
        o = retrieve_meshes.run(["--video", "clip", "--detections", "dets.json", "--topk", str(topk)] + common)
        assert o.name == f"props-ground-box-0.2-text-0.2-ffa-22-top-{topk}_clip.json"
        got = json.loads(o.read_text())
        assert len(got) == len(vprops) == N_FRAMES * 2
        per_obj = [(g_["mesh"], g_["score"]) for g_ in got[:2]]
        for i, g_ in enumerate(got):
            assert (g_["mesh"], g_["score"]) == per_obj[i % 2] and g_["image_id"] == i // 2 and g_["bbox"] == vprops[i]["bbox"]
        outs[topk] = got
    monkeypatch.setattr(retrieve_meshes, "WINDOW", 1)           # frame by frame: the same soft vote
    o1 = retrieve_meshes.run(["--video", "clip", "--detections", "dets.json", "--topk", "0", "--output", "props_w1.json"] + common)
    assert json.loads(o1.read_text()) == outs[0]
    monkeypatch.setattr(retrieve_meshes, "WINDOW", 8)
    # two ranks (frames sharded, soft vote as a collective) write the same file
    first = (vd / "props-ground-box-0.2-text-0.2-ffa-22-top-0_clip.json").read_text()
    _run_ranks("scripts.retrieve_meshes", ["--video", "clip", "--detections", "dets.json", "--topk", "0"] + common, root, 2, 29741)
    assert (vd / "props-ground-box-0.2-text-0.2-ffa-22-top-0_clip.json").read_text() == first
    for g_ in outs[0]:
        g_["scale"] = 0.1
    (vd / "props_retrieved.json").write_text(json.dumps(outs[0]))
    dino_inference_video.run(["--video", "clip", "--proposals", "props_retrieved.json", "--n_views", str(N_VIEWS), "--model", MODEL,
                              "--allow_random_weights", "--n_fine_poses", "2000"])
    dfv = pd.read_csv(root / "data" / "results" / "videos" / "clip" / "props_retrieved_dinopose_layer_22_bbext_0.05_depth_zoedepth.csv")
    assert len(dfv) == N_FRAMES * 2 and dfv["obj_id"].tolist()[:2] == [g_["mesh"] for g_ in outs[0][:2]]
