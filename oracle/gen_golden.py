"""Generate tests/golden/*.npz by IMPORTING the reference (/root/reference) in this container and running its
own functions on seeded inputs.  Only inputs/expected outputs (data) are written; no reference source travels.

    python -m oracle.gen_golden            # needs /root/reference; run once, fixtures are committed

Shims (sys.modules) stand in for packages the image lacks but whose code is NOT on the functions under test:
loguru (logger no-op), torchvision.transforms.Normalize, skimage, pyrender/trimesh/cv2 (import-time only), and a
bare `sam2` package whose __path__ points at the vendored sam2 so `sam2.utils.amg` imports without hydra.
`'cuda'` is mapped to `'cpu'` and DINOv2FeatureExtractor is replaced by a deterministic stand-in for the
estimator fixture — the ViT itself is un-vendored third-party code and is pinned separately
(oracle/vit_ref.py vs transformers' Dinov2WithRegistersModel).
"""
from __future__ import annotations

import hashlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None
    mod("loguru", logger=_Logger())

    class Normalize(torch.nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            self.mean, self.std = mean, std

        def forward(self, t):
            mean = torch.as_tensor(self.mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
            return (t - mean) / std
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Normalize=Normalize)
    sk = mod("skimage")
    sk.measure = mod("skimage.measure", regionprops=None)
    sk.morphology = mod("skimage.morphology", isotropic_erosion=None)
    pr = mod("pyrender", IntrinsicsCamera=lambda **kw: None, OffscreenRenderer=lambda *a, **kw: None)
    pr.constants = mod("pyrender.constants", RenderFlags=types.SimpleNamespace(SKIP_CULL_FACES=0))
    mod("trimesh", Trimesh=type("Trimesh", (), {}), PointCloud=type("PointCloud", (), {}))
    mod("cv2")
    sam2 = mod("sam2")
    sam2.__path__ = [str(REF / "segment-anything-2" / "sam2")]
    # the reference's `src` / `scripts` are namespace packages; this repo ships regular alias packages of the same
    # names, which would win regardless of path order — so drop the repo root from sys.path while importing.
    repo = str(Path(__file__).resolve().parent.parent)
    sys.path[:] = [str(REF)] + [p for p in sys.path if p not in ("", ".", repo) and Path(p or ".").resolve() != Path(repo)]
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.") or m == "scripts" or m.startswith("scripts.")]:
        del sys.modules[name]
    sys.meta_path[:] = [f for f in sys.meta_path if type(f).__name__ != "_LazyAlias"]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def main():
    assert REF.exists(), "/root/reference is only present in the build container"
    install_shims()
    OUT.mkdir(parents=True, exist_ok=True)
    from src.utils.bbox_utils import CropResizePad
    from src.pipeline.utils import Proposals, depthmap_to_pointcloud, get_z_from_pointcloud, mask_to_bbox
    from sam2.utils.amg import mask_to_rle_pytorch, rle_to_mask

    rng = np.random.Generator(np.random.PCG64(2024))

    # ---- a8: rotation grids ----------------------------------------------------------------------
    import src.pipeline.estimators.pose_estimator as pe_mod
    poses600 = np.array(pe_mod.DinoPoseEstimator.generate_poses(600))
    poses8 = np.array(pe_mod.DinoPoseEstimator.generate_poses(8))
    poses20k = np.array(pe_mod.DinoPoseEstimator.generate_poses(20000))
    np.savez_compressed(OUT / "poses.npz", poses600=poses600, poses8=poses8, poses20k_every100=poses20k[::100],
                        poses20k_sum=poses20k.sum(axis=0))

    # ---- a5: CropResizePad -----------------------------------------------------------------------
    H, W = 96, 128
    img = rng.random((3, H, W)).astype(np.float32)
    boxes = [[0, 0, W, H], [10, 10, 60, 60], [5, 7, 75, 47], [30, 20, 73, 63], [100, 70, 128, 96], [0, 0, 7, 5],
             [60, 1, 66, 95], [1, 40, 127, 47], [20, 20, 62, 62], [20, 20, 41, 41], [3, 3, 45, 24]]
    for _ in range(40):
        x0, y0 = int(rng.integers(0, W - 8)), int(rng.integers(0, H - 8))
        boxes.append([x0, y0, int(rng.integers(x0 + 4, W + 1)), int(rng.integers(y0 + 4, H + 1))])
    boxes = np.array(boxes, dtype=np.int32)
    crp = {}
    for target in (42, 30):
        for ext in (0, 0.05, 0.1, 0.2):
            proc = CropResizePad(target, (H, W), bbox_extend=ext)
            out = proc(torch.from_numpy(img)[None].repeat(len(boxes), 1, 1, 1), torch.from_numpy(boxes))
            crp[f"t{target}_e{ext}_first8"] = out.numpy()[:8]
            crp[f"t{target}_e{ext}_sha"] = np.array([sha(o) for o in out.numpy()])
    # full-size (420) results pinned by hash + samples: image 480x640, the quirky side lengths of App. A-10
    H2, W2 = 480, 640
    img2 = np.random.Generator(np.random.PCG64(7)).random((3, H2, W2)).astype(np.float32)
    boxes2 = np.array([[10, 10, 210, 210], [5, 7, 305, 207], [100, 50, 521, 471], [0, 0, 640, 480], [17, 33, 400, 90],
                       [50, 60, 210, 165], [300, 100, 405, 400], [0, 0, 600, 300]], dtype=np.int32)
    big = {}
    for ext in (0, 0.05, 0.1):
        out = CropResizePad(420, (H2, W2), bbox_extend=ext)(torch.from_numpy(img2)[None].repeat(len(boxes2), 1, 1, 1),
                                                           torch.from_numpy(boxes2)).numpy()
        big[f"sha_e{ext}"] = np.array([sha(o) for o in out])
        big[f"diag_e{ext}"] = out[:, :, np.arange(0, 420, 7), np.arange(0, 420, 7)]
    np.savez_compressed(OUT / "crop_resize_pad.npz", img=img, boxes=boxes, boxes2=boxes2, **crp, **big)

    # ---- a5: Proposals + RLE codec ----------------------------------------------------------------
    Hi, Wi = 120, 160
    image = rng.integers(0, 256, size=(Hi, Wi, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:Hi, 0:Wi]
    masks = np.stack([((yy - 40) / 25.0) ** 2 + ((xx - 50) / 35.0) ** 2 <= 1, (yy > 60) & (yy < 110) & (xx > 80) & (xx < 150),
                      ((yy - 70) / 40.0) ** 2 + ((xx - 100) / 20.0) ** 2 <= 1])
    pboxes = np.array([[15, 15, 86, 66], [80, 60, 150, 110], [80, 30, 121, 111]], dtype=np.int64)
    prop = {}
    for mask_rgb in (True, False):
        for ext in (0.05, 0.1, 0.2):
            p = Proposals(image, {"masks": torch.from_numpy(masks), "boxes": torch.from_numpy(pboxes)}, 56, 1, 2,
                          bbox_extend=ext, mask_rgb=mask_rgb)
            prop[f"props_rgb{int(mask_rgb)}_e{ext}"] = p.proposals.numpy()
            prop[f"pmask_rgb{int(mask_rgb)}_e{ext}"] = p.proposals_masks.numpy()
    rles = mask_to_rle_pytorch(torch.from_numpy(masks))
    back = np.stack([rle_to_mask(r) for r in rles])
    assert np.array_equal(back, masks)
    p.meshes, p.scores = ["a", "b", "c"], [0.5, 0.25, 0.125]
    bop = p.to_bop_dict()
    np.savez_compressed(OUT / "proposals.npz", image=image, masks=masks, boxes=pboxes, **prop,
                        rle_counts=np.array([np.array(r["counts"], dtype=np.int64) for r in rles], dtype=object),
                        rle_size=np.array(rles[0]["size"]), bop_bbox=np.array([b["bbox"] for b in bop]), allow_pickle=True)

    # ---- a9: depth -> point cloud -> z -------------------------------------------------------------
    K420 = np.array([[600, 0, 210], [0, 600, 210], [0, 0, 1]], dtype=np.int64)
    depth = np.zeros((3, 420, 420), np.float32)
    y2, x2 = np.mgrid[0:420, 0:420]
    depth[0][((y2 - 200) / 90.0) ** 2 + ((x2 - 220) / 120.0) ** 2 <= 1] = 1.05
    depth[1][(y2 > 100) & (y2 < 330) & (x2 > 150) & (x2 < 260)] = 0.9
    depth[1] += (depth[1] > 0) * (x2 / 4200.0).astype(np.float32)
    depth[2][200:203, 100:104] = 1.2
    Kq = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]], dtype=np.float32)
    bbox = np.array([200, 150, 330, 300])
    tco, ext_xy = [], []
    for i in range(3):
        pc = depthmap_to_pointcloud(depth[i], K420)
        ext_xy.append([pc[:, 0].max() - pc[:, 0].min(), pc[:, 1].max() - pc[:, 1].min(), len(pc)])
        mean = pc.mean(axis=0)
        pc2 = pc - mean
        pc2 /= 0.25
        pc2 *= 0.07
        pc2 += mean
        tco.append(get_z_from_pointcloud(bbox, pc2, Kq, poses600[5 + i]))
    bbs = np.stack([mask_to_bbox(depth[i] > 0) for i in range(3)])
    np.savez_compressed(OUT / "depth_pose.npz", depth=depth, K420=K420, Kq=Kq, bbox=bbox, est_scale=0.07,
                        tco=np.stack(tco), extents=np.array(ext_xy), init_pose_idx=np.array([5, 6, 7]), mask_bbox=bbs)

    # ---- a10: geodesic neighbourhood ---------------------------------------------------------------
    import src.pipeline.estimators.online_pose_estimator as on_mod
    geo = {}
    for j, qi in enumerate((0, 1234, 19999)):
        q = poses20k[qi].copy()
        if j == 1:  # perturb so the query is not a grid member
            from scipy.spatial.transform import Rotation as Rot
            q[:3, :3] = Rot.from_rotvec([0.02, -0.03, 0.05]).as_matrix() @ q[:3, :3]
        d = on_mod.DinoOnlinePoseEstimator.geodesic_distance(poses20k[:, :3, :3], q)
        geo[f"q{j}"] = q
        geo[f"close{j}"] = np.where(d < 15)[0]
        geo[f"dists{j}_every50"] = d[::50]
    np.savez_compressed(OUT / "geodesic.npz", **geo)

    # ---- a7: DinoPoseEstimator.forward with a deterministic stand-in extractor (CPU, bf16) ----------
    _orig_to = torch.Tensor.to

    def _to(self, *a, **kw):
        a = tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a)
        if kw.get("device") == "cuda":
            kw["device"] = "cpu"
        return _orig_to(self, *a, **kw)
    torch.Tensor.to = _to
    _orig_mto = torch.nn.Module.to
    torch.nn.Module.to = lambda self, *a, **kw: _orig_mto(self, *[("cpu" if x == "cuda" else x) for x in a], **kw)

    class FakeExtractor(torch.nn.Module):
        """deterministic stand-in: 6x6 average-pooled pixels -> fixed random projection to 64-d 'patch features'"""

        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(99)
            self.proj = torch.nn.Parameter(torch.randn(3 * 7 * 7, 64, generator=g), requires_grad=False)

        def forward(self, images, layer=22, feature_type="patch"):
            x = images.float()
            B = x.shape[0]
            pt = x.unfold(2, 7, 7).unfold(3, 7, 7)            # [B,3,6,6,7,7]
            pt = pt.permute(0, 2, 3, 1, 4, 5).reshape(B, 36, 147)
            return (pt @ self.proj.float()).to(torch.bfloat16)
    pe_mod.DINOv2FeatureExtractor = FakeExtractor
    import tempfile
    est = pe_mod.DinoPoseEstimator(n_poses=16, cache_size=0, cache_dir=tempfile.mkdtemp())
    T = 16
    tmpl = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).random((T, 3, 42, 42)).astype(np.float32))
    tdepth = torch.zeros(T, 420, 420)
    for t in range(T):
        tdepth[t, 120 + 3 * t:300 - 2 * t, 140 - t:280 + 2 * t] = 1.1 - 0.01 * t
    query = tmpl[11] * 0.9 + 0.1 * torch.from_numpy(np.random.Generator(np.random.PCG64(6)).random((3, 42, 42)).astype(np.float32))
    td = {"templates": tmpl, "depths": tdepth, "intrinsic": torch.from_numpy(K420), "model_name": "m"}
    out = est.forward(query, td, Kq, torch.tensor([200, 150, 330, 300]), 0.07, return_query_feat=True)
    feats_t = est.feature_extractor(tmpl.to(torch.bfloat16))
    torch.Tensor.to, torch.nn.Module.to = _orig_to, _orig_mto
    import shutil
    est.__class__.__del__ = lambda self: None
    shutil.rmtree(est.cache_dir, ignore_errors=True)
    # all 16 scores, recomputed exactly like pose_estimator.py:85-88 (so the full vector is pinned, not just top-3)
    import torch.nn.functional as F
    from einops import einsum
    sc_all = einsum(F.normalize(feats_t, dim=-1), F.normalize(out["query_feat"], dim=-1), "b n d, b n d -> b n").mean(dim=-1)
    np.savez_compressed(OUT / "pose_estimator.npz", templates=tmpl.numpy(), depths=tdepth.numpy(), query=query.numpy(),
                        Kq=Kq, bbox=np.array([200, 150, 330, 300]), est_scale=0.07, mesh_poses=np.array(est.mesh_poses),
                        tmpl_feats_bits=bf16_bits(feats_t), query_feat_bits=bf16_bits(out["query_feat"]),
                        scores_top3=np.asarray(out["scores"]), tco=np.stack(out["TCO"]),
                        scores_all=sc_all.float().numpy())

    # ---- a3/a4: masked-mean FFA + bank top-k evaluated with the reference's own torch expressions (CPU bf16) ----
    D, N = 1024, 3000
    bank = np.random.Generator(np.random.PCG64(21)).standard_normal((N, D)).astype(np.float32)
    bank += 2.0 * np.random.Generator(np.random.PCG64(22)).standard_normal(D).astype(np.float32)
    rf = F.normalize(torch.from_numpy(bank).to(torch.bfloat16), dim=-1)          # extract_proposals_ground.py:40-41
    feat = torch.from_numpy(np.random.Generator(np.random.PCG64(23)).standard_normal((2, 900, D)).astype(np.float32)).to(torch.bfloat16)
    m30 = np.random.Generator(np.random.PCG64(24)).random((2, 900)) < 0.35
    ffa = torch.stack([f[torch.from_numpy(m)].mean(dim=0) for f, m in zip(feat, m30)])  # :130-132
    q = F.normalize(ffa, dim=-1)                                                          # :134
    scores = [(rf @ qq).float() for qq in q]                                              # :137
    top = [torch.topk(s, 100) for s in scores]                                            # :140
    # big random inputs are regenerated from their PCG64 seeds by the test; outputs are pinned here
    np.savez_compressed(OUT / "retrieval.npz", seeds=np.array([21, 22, 23, 24]), N=N, D=D,
                        bank_norm_sha=sha(bf16_bits(rf)), bank_norm_rows0_8=bf16_bits(rf)[:8],
                        feat_sha=sha(bf16_bits(feat)), mask30=m30, ffa_bits=bf16_bits(ffa), q_bits=bf16_bits(q),
                        scores=np.stack([s.numpy() for s in scores]), top_scores=np.stack([t.values.numpy() for t in top]),
                        top_idx=np.stack([t.indices.numpy() for t in top]))
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")))


if __name__ == "__main__":
    main()
