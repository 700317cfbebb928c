"""Drop-in surface on the MI355X: the mirrored reference classes (Proposals, CropResizePad, MeshRenderer, the two
estimators, WebTemplateDataset, CLI helpers) against the reference's golden outputs and known answers."""
import io
import json
import tarfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _g(golden_dir, name):
    return np.load(golden_dir / name, allow_pickle=True)


@pytest.mark.parametrize("mask_rgb", [True, False])
@pytest.mark.parametrize("ext", [0.05, 0.1, 0.2])
def test_proposals_class_matches_reference(golden_dir, mask_rgb, ext):
    from src.pipeline.utils import Proposals          # reference import path, resolved by the alias package
    g = _g(golden_dir, "proposals.npz")
    p = Proposals(g["image"], {"masks": torch.from_numpy(g["masks"]), "boxes": torch.from_numpy(g["boxes"])}, 56, 1, 2,
                  bbox_extend=ext, mask_rgb=mask_rgb)
    assert np.array_equal(p.proposals.cpu().numpy(), g[f"props_rgb{int(mask_rgb)}_e{ext}"])
    assert np.array_equal(p.proposals_masks.cpu().numpy(), g[f"pmask_rgb{int(mask_rgb)}_e{ext}"])
    p.meshes, p.scores = ["a", "b", "c"], [0.5, 0.25, 0.125]
    bop = p.to_bop_dict()
    assert [b["bbox"] for b in bop] == g["bop_bbox"].tolist()
    assert [b["segmentation"]["counts"] for b in bop] == [list(c) for c in g["rle_counts"]]
    assert bop[0]["scene_id"] == 1 and bop[0]["image_id"] == 2 and bop[0]["time"] == 0.01
    json.dumps(bop)


@pytest.mark.parametrize("ext", [0, 0.05, 0.1])
def test_crop_resize_pad_class_matches_reference_420(golden_dir, ext):
    import hashlib
    from src.utils.bbox_utils import CropResizePad
    g = _g(golden_dir, "crop_resize_pad.npz")
    img2 = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).random((3, 480, 640)).astype(np.float32))
    out = CropResizePad(420, (480, 640), bbox_extend=ext)(img2[None], torch.from_numpy(g["boxes2"])).cpu().numpy()
    assert [hashlib.sha256(o.tobytes()).hexdigest() for o in out] == list(g[f"sha_e{ext}"])


class _StoredExtractor:
    """returns the golden stand-in features: [16,...] for the template batch, [1,...] for the query"""

    def __init__(self, g):
        from oracle import fp_oracle as fo
        self.t = fo.bits_to_torch(g["tmpl_feats_bits"]).cuda()
        self.q = fo.bits_to_torch(g["query_feat_bits"]).cuda()

    def __call__(self, images, layer=22, feature_type="patch"):
        return self.t[: images.shape[0]] if images.shape[0] > 1 else self.q


def test_pose_estimator_forward_matches_reference(golden_dir, tmp_path):
    from src.pipeline.estimators.pose_estimator import DinoPoseEstimator
    g = _g(golden_dir, "pose_estimator.npz")
    est = DinoPoseEstimator(n_poses=16, cache_size=2, cache_dir=tmp_path / "c", feature_extractor=_StoredExtractor(g))
    assert np.allclose(np.array(est.mesh_poses), g["mesh_poses"], atol=1e-15)
    td = {"templates": torch.from_numpy(g["templates"]), "depths": torch.from_numpy(g["depths"]), "model_name": "m",
          "intrinsic": torch.tensor([[600, 0, 210], [0, 600, 210], [0, 0, 1]])}
    for _ in range(2):  # second call hits the device LRU
        out = est.forward(torch.from_numpy(g["query"]), td, g["Kq"], torch.from_numpy(g["bbox"]), float(g["est_scale"]),
                          return_query_feat=True)
        assert np.array_equal(out["scores"], g["scores_top3"])               # bit-identical bf16 scores
        assert np.abs(out["TCO"][0] - g["tco"][0]).max() < 1e-6                 # untied winner: same pose
        assert set(out) >= {"TCO", "scores", "proposal", "K", "bbox", "retrieved_proposals", "query_feat"}
        assert len(out["TCO"]) == 3 and len(out["retrieved_proposals"]) == 3
    assert list(est.feature_cache) == ["m"]
    # all 16 scores through the public scorer
    # (the device store holds the rows PRE-NORMALISED, SURVEY §8 f-1: the streaming-dot scorer must reproduce the reference's
    #  normalise-every-call scores bit for bit)
    s = est.score_templates(est.feature_cache["m"], out["query_feat"], templates_normalized=True).cpu().numpy()
    assert np.array_equal(s, g["scores_all"])
    # ---- forward_many (the proposals of one image in one step, one device -> host copy) == forward per item, exactly
    td2 = dict(td, model_name="m2", depths=torch.from_numpy(g["depths"]).cuda())          # device-resident depths, as the template loader returns them
    items = [dict(proposal=torch.from_numpy(g["query"]), template_dict=lambda: td, K=g["Kq"], bbox=torch.from_numpy(g["bbox"]), est_scale=float(g["est_scale"])),
             dict(proposal=torch.from_numpy(g["query"]), template_dict=td2, K=g["Kq"], bbox=torch.from_numpy(g["bbox"]), est_scale=0.5 * float(g["est_scale"]))]
    many = est.forward_many(items, return_query_feat=True)
    one = [est.forward(torch.from_numpy(g["query"]), t_, g["Kq"], torch.from_numpy(g["bbox"]), sc_) for t_, sc_ in ((td, float(g["est_scale"])), (td2, 0.5 * float(g["est_scale"])))]
    for a_, b_ in zip(many, one):
        assert np.array_equal(a_["scores"], b_["scores"]) and a_["scores"].dtype == b_["scores"].dtype == np.float32
        assert all(np.array_equal(x, y) for x, y in zip(a_["TCO"], b_["TCO"])) and len(a_["retrieved_proposals"]) == 3
    assert np.array_equal(many[0]["scores"], g["scores_top3"]) and "query_feat" in many[0] and est.forward_many([]) == []
    # ---- the reference's cache semantics (pose_estimator.py:43-53,63-65): an entry the LRU evicts is written to <cache_dir> (under
    # flock) and a revisit reads it back instead of running the ViT again; save_all keeps the RAW features as <name>.pth
    class Counting(_StoredExtractor):
        calls = 0

        def __call__(self, images, layer=22, feature_type="patch"):
            Counting.calls += images.shape[0] > 1
            return super().__call__(images, layer, feature_type)
    est1 = DinoPoseEstimator(n_poses=16, cache_size=1, save_all=True, cache_dir=tmp_path / "c1", feature_extractor=Counting(g))
    args = (torch.from_numpy(g["query"]), td, g["Kq"], torch.from_numpy(g["bbox"]), float(g["est_scale"]))
    assert np.array_equal(est1.forward(*args)["scores"], g["scores_top3"]) and Counting.calls == 1
    raw = torch.load(tmp_path / "c1" / "m.pth")
    assert raw.dtype == torch.bfloat16 and raw.device.type == "cpu" and torch.equal(raw.cuda(), Counting(g).t[:16])   # the reference's file format
    est1.forward(torch.from_numpy(g["query"]), dict(td, model_name="other"), g["Kq"], torch.from_numpy(g["bbox"]), float(g["est_scale"]))
    assert list(est1.feature_cache) == ["other"] and (est1._spill_dir / "m.evicted.pth").exists() and Counting.calls == 2
    assert est1._spill_dir.parent == tmp_path / "c1" and not list((tmp_path / "c1").glob("*.tmp"))
    out1 = est1.forward(*args)                                       # "m" comes back from the spill file: no third template pass
    assert Counting.calls == 2 and list(est1.feature_cache) == ["m"] and np.array_equal(out1["scores"], g["scores_top3"])
    assert (est1._spill_dir / "other.evicted.pth").exists()
    # a spill file is never another run's input: rows spilled at one layer are not served at another, and the directory
    # goes with the estimator even under save_all (the raw <name>.pth files stay)
    blob = torch.load(est1._spill_dir / "other.evicted.pth")
    assert blob["layer"] == 22 and blob["features_normalized"].shape[0] == 16
    spill = est1._spill_dir
    del est1
    import gc
    gc.collect()
    assert not spill.exists() and (tmp_path / "c1" / "m.pth").exists()


def _mesh():
    import bench
    from freepose_amd.mesh_io import TriMesh
    v, f, c = bench.synthetic_mesh(4)
    return TriMesh(v, f, c)


def test_mesh_renderer_api_and_generate_proposals():
    from src.pipeline.retrieval.renderer import MeshRenderer
    r = MeshRenderer(6)
    assert len(r.mesh_poses) == 6 and np.allclose(r.mesh_poses[0][:3, 3], [0, 0, 1.1])
    mesh = _mesh()
    res = r.render(mesh, scale=0.25)
    rgb, depth, R = res[2]
    assert rgb.shape == (420, 420, 3) and rgb.dtype == np.uint8 and depth.shape == (420, 420) and R.shape == (3, 3)
    assert (depth > 0).sum() > 5000 and depth[depth > 0].min() > 0.8 and depth.max() < 1.4
    crops, poses, masks = MeshRenderer.generate_proposals(res)
    assert crops.shape == (6, 3, 420, 420) and masks.shape == (6, 420, 420) and len(poses) == 6
    # the reference-style list-of-tuples input goes through the same kernels
    crops2, _, masks2 = MeshRenderer.generate_proposals([res[i] for i in range(6)])
    assert torch.equal(crops, crops2) and torch.equal(masks, masks2)
    # crop of a render == oracle crop of the same render with its mask bbox
    from oracle import fp_oracle as fo
    bb = MeshRenderer.mask_to_bbox(depth > 0)
    ref = fo.crop_resize_pad(rgb[None], bb[None].astype(np.int32), 420, 0.0)
    assert np.array_equal(crops[2].cpu().numpy(), ref[0])


def test_online_estimator_recovers_planted_pose(tmp_path):
    """known answer: the query IS a render of fine-grid pose i; starting from a neighbouring pose the render-and-compare
    step must select i with score ~1, and the recovered z must match the planted scale (2 deg / 2 mm budget)."""
    from src.pipeline.estimators.online_pose_estimator import DinoOnlinePoseEstimator
    from src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    from src.pipeline.retrieval.renderer import MeshRenderer
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fe = DINOv2FeatureExtractor("dinov2_vits14_reg", seed=4)
    est = DinoOnlinePoseEstimator(n_coarse_poses=8, n_fine_poses=20000, cache_size=0, cache_dir=tmp_path / "c", feature_extractor=fe)
    mesh = _mesh()
    i = 7777
    true_pose = est.fine_mesh_poses[i]
    render = est.renderer.render_from_poses(mesh, [true_pose], scale=0.25)
    crops, _, masks, ext = MeshRenderer.generate_proposals(render, return_extents=True)
    d = DinoOnlinePoseEstimator.geodesic_distance(est.fine_mesh_poses[:, :3, :3], true_pose)
    nb = np.argsort(d)[3]                       # a different grid pose, a few degrees away
    assert 0 < d[nb] < 15
    K = np.array([[600.0, 0, 210], [0, 600.0, 210], [0, 0, 1]])
    e = ext[0].cpu().numpy()
    bbox = torch.tensor([int(e[0]), int(e[1]), int(e[2]), int(e[3])])
    out = est.forward(crops[0].float(), masks[0], None, mesh, K, bbox, est_scale=0.25, prev_pose=est.fine_mesh_poses[nb])
    R = out["TCO"][0][:3, :3]
    ang = np.degrees(np.arccos(np.clip((np.trace(R @ true_pose[:3, :3].T) - 1) / 2, -1, 1)))
    assert ang < 1e-6, f"selected a pose {ang} deg away"
    assert float(out["scores"][0]) > 0.99
    assert abs(out["TCO"][0][2, 3] - 1.1) < 0.03          # z from the silhouette extents (1 px ~ 0.5 % of z)
    # mask-weighted scoring variant runs and still finds the pose
    out2 = est.forward_fine(crops[0].float(), masks[0], None, mesh, K, bbox, 0.25, est.fine_mesh_poses[nb], mask_scores=True)
    assert np.allclose(out2["TCO"][0][:3, :3], true_pose[:3, :3])
    # the objects of a frame in ONE batched step (scripts.dino_inference_video): three items — another planted pose, a second
    # mesh, the mask-free first item again — give exactly the results of three separate steps
    from tests._meshes import textured_cube
    from freepose_amd.mesh_io import TriMesh
    v2, f2, _ = textured_cube()
    mesh2 = TriMesh(v2 * np.array([1.0, 0.6, 0.4], np.float32), f2, np.random.default_rng(3).integers(0, 255, size=(len(v2), 3), dtype=np.uint8))
    items = []
    for (msh, j, k) in ((mesh, 7777, 3), (mesh2, 1234, 2), (mesh, 15000, 4)):
        tp = est.fine_mesh_poses[j]
        rj = est.renderer.render_from_poses(msh, [tp], scale=0.25)
        cj, _, mj, ej = MeshRenderer.generate_proposals(rj, return_extents=True)
        dj = DinoOnlinePoseEstimator.geodesic_distance(est.fine_mesh_poses[:, :3, :3], tp)
        ee = ej[0].cpu().numpy()
        items.append(dict(proposal=cj[0].float(), proposal_mask=mj[0], template_dict=None, mesh=msh, K=K,
                          bbox=torch.tensor([int(ee[0]), int(ee[1]), int(ee[2]), int(ee[3])]), est_scale=0.25,
                          prev_pose=est.fine_mesh_poses[np.argsort(dj)[k]]))
    many = est.forward_fine_many(items)
    for it, got in zip(items, many):
        one = est.forward_fine(it["proposal"], it["proposal_mask"], None, it["mesh"], K, it["bbox"], 0.25, it["prev_pose"])
        assert np.array_equal(got["TCO"][0], one["TCO"][0]) and got["scores"][0] == one["scores"][0]
    assert np.allclose(many[1]["TCO"][0][:3, :3], est.fine_mesh_poses[1234][:3, :3])


def test_hypothesis_store_gives_the_recomputed_results(tmp_path):
    """DinoOnlinePoseEstimator keeps the features / extents / masks of fine-grid hypotheses per mesh between frames and only renders
    those that ENTER the neighbourhood; the reference recomputes every hypothesis in every frame.  Over a clip — two objects that share
    a mesh plus one with its own, plain and mask-weighted scores, a store so small that it overflows and is emptied, and no store at
    all — every pose and every score must be identical, and the store must really save ViT work."""
    import warnings
    from src.pipeline.estimators.online_pose_estimator import DinoOnlinePoseEstimator
    from src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    from src.pipeline.retrieval.renderer import MeshRenderer
    from scipy.spatial.transform import Rotation as Rot
    from tests._meshes import textured_cube
    from freepose_amd.mesh_io import TriMesh
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fe = DINOv2FeatureExtractor("dinov2_vits14_reg", seed=4)
    mesh = _mesh()
    v2, f2, _ = textured_cube()
    mesh2 = TriMesh(v2 * np.array([1.0, 0.6, 0.4], np.float32), f2, np.random.default_rng(3).integers(0, 255, size=(len(v2), 3), dtype=np.uint8))
    K = np.array([[600.0, 0, 210], [0, 600.0, 210], [0, 0, 1]])
    n_frames = 14
    calls = {}

    def run(cap, mask_scores):
        est = DinoOnlinePoseEstimator(n_coarse_poses=8, n_fine_poses=20000, cache_size=0, cache_dir=tmp_path / f"c{cap}", feature_extractor=fe,
                                      hypothesis_cache=cap)
        crops_seen = [0]
        inner = est.feature_extractor

        class Counting(torch.nn.Module):
            def forward(self, x, **kw):
                crops_seen[0] += x.shape[0]
                return inner(x, **kw)
        est.feature_extractor = Counting()
        starts = [(mesh, 7777, np.array([0.3, 1.0, 0.2])), (mesh2, 1234, np.array([1.0, -0.2, 0.4])), (mesh, 7790, np.array([-0.5, 0.3, 1.0]))]
        prev = [est.fine_mesh_poses[j] for _, j, _ in starts]
        out_all = []
        for fr in range(n_frames):
            items = []
            for o, (msh, j, ax) in enumerate(starts):
                tp = np.eye(4)
                tp[:3, :3] = Rot.from_rotvec(np.deg2rad(2.5 * fr) * ax / np.linalg.norm(ax)).as_matrix() @ est.fine_mesh_poses[j][:3, :3]
                tp[:3, 3] = est.fine_mesh_poses[j][:3, 3]
                rj = est.renderer.render_from_poses(msh, [tp], scale=0.25)
                cj, _, mj, ej = MeshRenderer.generate_proposals(rj, return_extents=True)
                ee = ej[0].cpu().numpy()
                items.append(dict(proposal=cj[0].float(), proposal_mask=mj[0], template_dict=None, mesh=msh, K=K,
                                  bbox=torch.tensor([int(ee[0]), int(ee[1]), int(ee[2]), int(ee[3])]), est_scale=0.25, prev_pose=prev[o]))
            outs = est.forward_fine_many(items, mask_scores=mask_scores)
            prev = [o_["TCO"][0] for o_ in outs]
            out_all.append([(o_["TCO"][0].copy(), float(o_["scores"][0])) for o_ in outs])
        calls[(cap, mask_scores)] = crops_seen[0]
        return out_all

    for mask_scores in (False, True):
        ref = run(0, mask_scores)                       # every hypothesis recomputed in every frame (the reference's behaviour)
        for cap in (768, 60):                           # a roomy store; one that overflows (three neighbourhoods of ~20 barely fit) and is emptied
            got = run(cap, mask_scores)
            for fr in range(n_frames):
                for (Ta, sa), (Tb, sb) in zip(ref[fr], got[fr]):
                    assert np.array_equal(Ta, Tb) and sa == sb, (mask_scores, cap, fr)
    # the object moves 2.5 degrees per frame through a ~9-degree grid: most neighbours repeat from frame to frame
    assert calls[(768, False)] < 0.45 * calls[(0, False)], calls
    assert calls[(60, False)] <= calls[(0, False)]
    # the poses follow the planted rotation (sanity of the scenario itself)
    last = ref[-1][0][0]
    assert np.isfinite(last).all() and abs(np.linalg.det(last[:3, :3]) - 1) < 1e-6


def test_more_tracked_meshes_than_hypothesis_stores(tmp_path):
    """a frame that tracks more distinct meshes than `hypothesis_meshes` (the store of the first object was evicted while the
    step still held it -> KeyError): every object keeps its store for the step, results == no store, and the LRU bound holds again once
    a later frame tracks fewer"""
    import warnings
    from src.pipeline.estimators.online_pose_estimator import DinoOnlinePoseEstimator
    from src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    from src.pipeline.retrieval.renderer import MeshRenderer
    from freepose_amd.mesh_io import TriMesh
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fe = DINOv2FeatureExtractor("dinov2_vits14_reg", seed=4)
    base = _mesh()
    rng = np.random.default_rng(8)
    meshes = [TriMesh(np.asarray(base.vertices, np.float32) * rng.uniform(0.5, 1.0, size=3).astype(np.float32), np.asarray(base.faces),
                      rng.integers(0, 255, size=(len(base.vertices), 3), dtype=np.uint8)) for _ in range(5)]
    K = np.array([[600.0, 0, 210], [0, 600.0, 210], [0, 0, 1]])

    def run(cap, n_stores):
        est = DinoOnlinePoseEstimator(n_coarse_poses=8, n_fine_poses=20000, cache_size=0, cache_dir=tmp_path / f"m{cap}", feature_extractor=fe,
                                      hypothesis_cache=cap, hypothesis_meshes=n_stores)
        out = []
        for fr, use in enumerate(([0, 1, 2, 3, 4], [4, 3, 2, 1, 0], [2, 0])):
            items = []
            for o in use:
                j = 500 + 37 * o
                rj = est.renderer.render_from_poses(meshes[o], [est.fine_mesh_poses[j]], scale=0.25)
                cj, _, mj, ej = MeshRenderer.generate_proposals(rj, return_extents=True)
                ee = ej[0].cpu().numpy()
                items.append(dict(proposal=cj[0].float(), proposal_mask=mj[0], template_dict=None, mesh=meshes[o], K=K,
                                  bbox=torch.tensor([int(ee[0]), int(ee[1]), int(ee[2]), int(ee[3])]), est_scale=0.25,
                                  prev_pose=est.fine_mesh_poses[j + fr]))
            out.append([(o_["TCO"][0].copy(), float(o_["scores"][0])) for o_ in est.forward_fine_many(items)])
            assert len(est._hyp_stores) <= max(n_stores, len(use)) or cap == 0
        return out, est

    ref, _ = run(0, 2)
    got, est = run(768, 2)
    for a, b in zip(ref, got):
        for (Ta, sa), (Tb, sb) in zip(a, b):
            assert np.array_equal(Ta, Tb) and sa == sb
    assert len(est._hyp_stores) == 2                  # the last frame tracked two meshes: back within the bound


def _png(arr, mode):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr, mode).save(b, format="PNG")
    return b.getvalue()


def test_template_dataset_and_bank_build(tmp_path):
    """synthetic shard in the reference's on-disk format -> WebTemplateDataset -> per-view FFA descriptors
    (extract_retrieval_features) -> merge -> TemplateBank retrieval finds the mesh itself"""
    from src.pipeline.retrieval.renderer import MeshRenderer
    from src.dataloader.template import WebTemplateDataset
    from freepose_amd.scripts.extract_retrieval_features import mesh_descriptors
    from freepose_amd.retrieval import TemplateBank
    from src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    import warnings
    n_views, names = 5, ["meshA", "meshB"]
    r = MeshRenderer(n_views)
    import bench
    from freepose_amd.mesh_io import TriMesh
    shard = tmp_path / "shards"
    shard.mkdir()
    with tarfile.open(shard / "shard-000000.tar", "w") as tar:
        for k, name in enumerate(names):
            v, f, c = bench.synthetic_mesh(3, seed=50 + k)
            v = v * ([1.0, 0.6, 0.8] if k else [0.7, 1.0, 0.5])
            res = r.render(TriMesh(v, f, c), scale=0.25)
            for j in range(n_views):
                rgb, depth, _ = res[j]
                for suffix, data in ((f"{name}_{j}.rgb.png", _png(rgb, "RGB")),
                                     (f"{name}_{j}.depth.png", _png((depth * 1000).astype(np.uint16), "I;16"))):
                    ti = tarfile.TarInfo(suffix)
                    ti.size = len(data)
                    tar.addfile(ti, io.BytesIO(data))
    (tmp_path / "list.csv").write_text("model_name\n" + "\n".join(names) + "\n")
    ds = WebTemplateDataset(str(shard), str(tmp_path / "list.csv"), crop=False, n_views=n_views)
    assert len(ds) == 2
    s = ds.get_template_by_name("meshB")
    assert s["templates"].shape == (n_views, 3, 420, 420) and s["masks"].shape == (n_views, 420, 420)
    assert s["model_name"] == "meshB" and s["tar_file"] == "shard-000000.tar" and s["depths"].dtype == torch.float32
    assert (shard / "shard-000000.npy").exists()                              # member index cached beside the tar
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = DINOv2FeatureExtractor("dinov2_vits14_reg", seed=6)
    rows = []
    for idx in range(2):
        d = mesh_descriptors(model, ds[idx], "ffa", 22, 4)
        assert d.shape == (n_views, 384) and d.dtype == np.float32 and np.isfinite(d).all()
        rows.append(d.mean(axis=0))
    bank = TemplateBank(np.stack(rows), names)
    # each mesh's own object descriptor (mean over its views, merge_features.py:31-33) must score highest against its own
    # bank row.  (With a random-init ViT-S two bright blobs give descriptors whose cosine rounds to the same bf16 score, so
    # the retrieved NAME is decided by the index tie rule; this is a plumbing test, not a discriminability test.)
    from freepose_amd import ops
    q = torch.from_numpy(np.stack(rows))
    sc, ix = bank.topk(ops.l2_normalize(q.to(torch.bfloat16)), 2)
    sc, ix = sc.cpu().numpy(), ix.cpu().numpy()
    for r in range(2):
        assert sorted(ix[r].tolist()) == [0, 1] and sc[r][ix[r] == r][0] >= sc[r].max() - 2 ** -7 and sc[r].max() > 0.99
    got, score, idx = bank.retrieve(ops.l2_normalize(q.to(torch.bfloat16)))
    # retrieve() is the top-1 of the same scan (own row or, inside the 2^-7 window asserted above, the other one by the tie rule)
    assert list(got) == [names[int(ix[r][0])] for r in range(2)] and all(g in names for g in got)
    # crop=True path (what the inference drivers use)
    ds2 = WebTemplateDataset(str(shard), str(tmp_path / "list.csv"), bbox_extend=0.05, n_views=n_views)
    s2 = ds2[0]
    assert s2["templates"].shape == (n_views, 3, 420, 420) and s2["intrinsic"].tolist() == [[600, 0, 210], [0, 600, 210], [0, 0, 1]]


def test_hot_path_batch_and_cli_rows(tmp_path):
    import bench
    from freepose_amd import ops
    from freepose_amd.pipeline import HotPath, pack_results
    from freepose_amd.retrieval import TemplateBank
    vit = ops.ViT("dinov2_vits14_reg", seed=2)
    bank = TemplateBank(bench.synthetic_bank(500, 384, seed=3))
    v, f, c = bench.synthetic_mesh(3)
    hp = HotPath(vit, bank, ops.Mesh(v, f, c), n_hyp=12, crop_res=224, k=20, vit_batch=5)
    crops, masks, K, boxes, scales = bench.synthetic_proposals(3, 224, seed=5)
    r1 = hp.run(crops.cuda(), masks.cuda(), K, boxes, scales)
    r2 = hp.run(crops[1:2].cuda(), masks[1:2].cuda(), K, boxes[1:2], scales[1:2])
    assert len(r1) == 3 and np.array_equal(r1[1].topk_idx, r2[0].topk_idx) and np.array_equal(r1[1].hyp_idx, r2[0].hyp_idx)
    assert np.allclose(r1[1].TCO[0], r2[0].TCO[0])
    rows = pack_results(r1)
    assert rows.shape == (3, 16) and torch.isfinite(rows).all()
    assert all(np.all(np.diff(r.topk_scores) <= 0) for r in r1)


def test_render_templates_cli_roundtrip(tmp_path, monkeypatch):
    """scripts.render_templates (HIP rasteriser) -> shard tar in the reference's layout -> WebTemplateDataset: the decoded
    views must equal a direct render (rgb exactly, depth to the u16-millimetre truncation of the format)."""
    import bench
    from freepose_amd.mesh_io import load_obj
    from freepose_amd.src.dataloader.template import WebTemplateDataset
    from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer
    from scripts import render_templates
    n_views = 12
    ids = ["mesh_a", "mesh_b"]
    for k, mid in enumerate(ids):
        v, f, c = bench.synthetic_mesh(2 + k)
        d = tmp_path / "mesh_cache" / mid
        d.mkdir(parents=True)
        with open(d / f"{mid}.obj", "w") as fh:
            for p, col in zip(v, c):
                fh.write(f"v {p[0]:.7f} {p[1]:.7f} {p[2]:.7f} {col[0] / 255:.6f} {col[1] / 255:.6f} {col[2] / 255:.6f}\n")
            for t in f:
                fh.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
    (tmp_path / "list.txt").write_text("\n".join(ids) + "\n")
    monkeypatch.delenv("SLURM_ARRAY_TASK_ID", raising=False)
    tar_path = render_templates.run(["--filelist", str(tmp_path / "list.txt"), "--mesh_root", str(tmp_path / "mesh_cache"),
                                     "--datasets_root", str(tmp_path / "datasets"), "--shards_folder", "sh",
                                     "--n_views", str(n_views)])
    assert tar_path.name == "shard-000000.tar" and tar_path.exists()
    with tarfile.open(tar_path) as tar:
        names = tar.getnames()
    assert len(names) == 2 * 2 * n_views and "mesha_0.rgb.png" in names and "meshb_11.depth.png" in names
    (tmp_path / "list.csv").write_text("model_name\n" + "\n".join(ids) + "\n")
    ds = WebTemplateDataset(str(tmp_path / "datasets" / "sh"), str(tmp_path / "list.csv"), crop=False, n_views=n_views)
    s = ds[1]
    assert s["model_name"] == "meshb" and s["depths"].shape == (n_views, 420, 420)
    mesh = load_obj(tmp_path / "mesh_cache" / "mesh_b" / "mesh_b.obj")
    mesh.apply_scale(0.25)
    direct = MeshRenderer(n_views).render(mesh)
    assert torch.equal(s["templates"].cpu(), ops_crop_identity(direct.rgb).cpu())
    d_direct = (direct.depth.cpu().numpy() * 1000).astype(np.uint16).astype(np.float32) / 1000
    assert np.array_equal(s["depths"].cpu().numpy(), (d_direct.astype(np.float64)).astype(np.float32)) or \
        np.abs(s["depths"].cpu().numpy() - d_direct).max() < 1e-6


def ops_crop_identity(rgb_u8):
    """what WebTemplateDataset(crop=False) returns for a batch of renders: the full frame through the crop op"""
    from freepose_amd import ops
    n, h, w = rgb_u8.shape[0], rgb_u8.shape[1], rgb_u8.shape[2]
    return ops.crop_resize_pad(rgb_u8, torch.tensor([[0, 0, w, h]] * n, dtype=torch.int32), h, 0.0)


def test_dino_inference_rows_batched_equals_one_by_one(tmp_path):
    """scripts.dino_inference.proposal_rows computes the query features of all proposals of an image in one ViT batch; the
    rows (scores, R, t) must equal running the estimator proposal by proposal, and keep the reference's CSV fields"""
    import warnings

    from freepose_amd.scripts.dino_inference import proposal_rows
    from freepose_amd.src.pipeline.utils import Proposals, mask_to_rle_pytorch
    from src.pipeline.estimators.pose_estimator import DinoPoseEstimator
    from src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    from src.pipeline.retrieval.renderer import MeshRenderer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fe = DINOv2FeatureExtractor("dinov2_vits14_reg", seed=6)
    est = DinoPoseEstimator(n_poses=12, cache_size=4, cache_dir=tmp_path / "c", feature_extractor=fe)
    renders = MeshRenderer(12).render(_mesh(), scale=0.25)
    crops, _, _ = MeshRenderer.generate_proposals(renders)
    td = {"templates": crops.float(), "depths": renders.depth, "model_name": "obj_1",
          "intrinsic": torch.tensor([[600, 0, 210], [0, 600, 210], [0, 0, 1]])}

    class _Templates:
        def get_template_by_name(self, name):
            return dict(td, model_name=name)

    rng = np.random.default_rng(3)
    image = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    props = []
    for (x, y, w, h) in [(100, 80, 150, 170), (300, 200, 120, 90), (20, 250, 200, 180)]:
        m = np.zeros((480, 640), dtype=np.uint8)
        yy, xx = np.mgrid[0:480, 0:640]
        m[((xx - (x + w / 2)) / (w / 2)) ** 2 + ((yy - (y + h / 2)) / (h / 2)) ** 2 <= 1] = 1
        props.append({"segmentation": mask_to_rle_pytorch(torch.from_numpy(m[None]))[0], "bbox": [x, y, w, h], "mesh": "obj_1"})
    K = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]])
    scales = [0.07, 0.1, 0.12]
    rows = proposal_rows(est, _Templates(), image, K, 48, 1, props, scales, 22, 128, 0.05)
    assert len(rows) == 3 and set(rows[0]) == {"scene_id", "im_id", "obj_id", "score", "R", "t", "bbox_visib", "scale", "time"}
    # one by one, without the batched query features
    from freepose_amd.src.pipeline.utils import rle_to_mask
    mk = torch.from_numpy(np.stack([rle_to_mask(p["segmentation"]) for p in props]))
    bx = torch.from_numpy(np.stack([np.array(p["bbox"]) for p in props]))
    bx[:, 2:] += bx[:, :2]
    single = Proposals(image, {"boxes": bx, "masks": mk}, 420, bbox_extend=0.05)
    for i, prop in enumerate(single.proposals):
        out = est(prop, td, K, bx[i], scales[i], layer=22, batch_size=128)
        assert float(out["scores"][0]) == float(rows[i]["score"])
        assert " ".join(str(x) for x in out["TCO"][0][:3, :3].flatten().tolist()) == rows[i]["R"]
        assert " ".join(str(x * 1000.0) for x in out["TCO"][0][:3, 3].tolist()) == rows[i]["t"]
