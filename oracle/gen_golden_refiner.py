"""tests/golden/refiner.npz: the TrackingRefiner host arithmetic computed by the REFERENCE's own code,
imported here from /root/reference.  Shimmed (import-time only, not on the functions under test): the packages
gen_golden.py shims plus open3d, torchvision.transforms.{ToTensor,Compose}, and `torchvision.ops.roi_align`, which is
replaced by a recorder returning zeros (RoIAlign itself is torchvision code, not the reference's; its restatement is held
by known answers — DESIGN.md §5).

    python -m oracle.gen_golden_refiner
"""
from __future__ import annotations

import sys
import types

import numpy as np
import torch

from oracle.gen_golden import OUT, REF, install_shims


def main():
    assert REF.exists(), "/root/reference is only present in the build container"
    install_shims()
    tv = sys.modules["torchvision"]
    tr = tv.transforms

    class ToTensor:
        def __call__(self, pic):
            a = np.asarray(pic)
            t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)
            return t.float().div(255) if t.dtype == torch.uint8 else t.float()

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    tr.ToTensor, tr.Compose = ToTensor, Compose
    recorded = {}

    def roi_align(image, boxes, output_size, sampling_ratio=-1, **kw):
        recorded["rois"] = boxes.clone()
        recorded["args"] = (tuple(output_size), sampling_ratio)
        return torch.zeros((boxes.shape[0], image.shape[1]) + tuple(output_size))
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.roi_align = roi_align
    sys.modules["torchvision.ops"] = tv.ops
    for name in ("open3d",):
        sys.modules[name] = types.ModuleType(name)
    cv2 = sys.modules["cv2"]
    cv2.INTER_CUBIC = 2

    from src.pipeline import refiner_utils
    from src.pipeline.estimators.tracking_refiner import TrackingRefiner

    rng = np.random.Generator(np.random.PCG64(77))
    verts = (rng.standard_normal((5000, 3)) * np.array([0.08, 0.05, 0.11])).astype(np.float64)
    mesh = types.SimpleNamespace(vertices=verts)
    tr_obj = TrackingRefiner.__new__(TrackingRefiner)
    K = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]])
    cases = []
    for i in range(6):
        ang = rng.standard_normal(3)
        th = np.linalg.norm(ang)
        k = ang / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = rng.uniform([-0.15, -0.1, 0.45], [0.15, 0.1, 1.4])
        image = rng.random((480, 640, 3)).astype(np.float32)
        crop, bbox, newK = tr_obj._crop_image(mesh, torch.from_numpy(image).permute(2, 0, 1).contiguous(), K, T)
        cases.append((T, bbox.numpy(), newK.numpy(), recorded["rois"].numpy()))
        assert recorded["args"] == ((518, 518), 2) and tuple(crop.shape) == (3, 518, 518)
    # the 100 sampled object points (np.random.seed(42); np.random.choice): homogeneous float32
    np.random.seed(42)
    pick = np.random.choice(np.arange(len(verts)), 100)
    sims = rng.random((7, 37, 37)).astype(np.float32) * (rng.random((7, 37, 37)) > 0.3)
    thr = [float(tr_obj._get_threshold_for_confidence(sims, top_quantile=q)) for q in (0.2, 0.05, 0.5)]
    direct_K = refiner_utils.update_K_with_crop(torch.from_numpy(K).float(), torch.tensor([[10.5, 20.25, 400.0, 300.75],
                                                                                            [-30.0, -12.0, 700.0, 520.0]]), 518, 518)
    np.savez_compressed(OUT / "refiner.npz", verts=verts, K=K, transforms=np.stack([c[0] for c in cases]),
                        bboxes=np.stack([c[1] for c in cases]), new_K=np.stack([c[2] for c in cases]),
                        rois=np.stack([c[3] for c in cases]), pick=pick, sims=sims, thresholds=np.array(thr),
                        direct_boxes=np.array([[10.5, 20.25, 400.0, 300.75], [-30.0, -12.0, 700.0, 520.0]], dtype=np.float32),
                        direct_new_K=direct_K.numpy())
    print("wrote", OUT / "refiner.npz", "thresholds", thr)


if __name__ == "__main__":
    main()
