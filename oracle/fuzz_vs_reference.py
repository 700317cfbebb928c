"""Randomised pinning of the oracle: IMPORT the reference (/root/reference) in this container and run its own functions beside the
oracle's restatements on thousands of seeded random inputs — the same comparison the golden fixtures make (oracle/gen_golden.py) at
inputs nobody chose.  Test infrastructure, run by hand here (the reference does not travel); its log is committed:

    python -m oracle.fuzz_vs_reference [cases-per-section] [seed]  >  profiles/r05_oracle_fuzz_vs_reference.log

Sections (reference function -> oracle / host restatement, criterion):
  crop      src/utils/bbox_utils.py CropResizePad.__call__           -> fpo_crop_resize_pad                       every byte equal; a box torch
            refuses (F.interpolate to 0 px) is refused by the oracle and by the mirror's host check (bbox_utils.unresizable_box), and no other
  proposals src/pipeline/utils.py Proposals (crops + crop masks)     -> fpo_crop_resize_pad modes 0 / 1 / 2       every byte equal
  rle       sam2/utils/amg.py mask_to_rle_pytorch / rle_to_mask      -> freepose_amd.src.pipeline.utils           equal
  depth     depthmap_to_pointcloud + get_z_from_pointcloud, mask_to_bbox -> fpo_depth_extents + z_from_extents    count / bbox equal, extents <= 1e-12, pose <= 1e-9 m
  poses     DinoPoseEstimator.generate_poses(n)                      -> fpo_generate_rotations                    <= 1e-15
  geodesic  DinoOnlinePoseEstimator.geodesic_distance < threshold    -> fpo_geodesic_select                       the same index list
  csv       scripts/dino_inference.py:113-127, dino_inference_video.py:160-176 (exec'd) -> scripts.dino_inference.pose_row   the CSV text
  score     pose_estimator.py:85-88 (normalize, einsum, mean in bf16 torch)   -> fpo_template_score                       <= 1 bf16 ulp, same arg-max when decisive
  retrieval extract_proposals_ground.py:40-41,130-140 (torch expressions)     -> fpo_bank_prepare / scores / topk, fpo_ffa  >= 99.8 % of scores identical, same top-100 multiset
  refiner   TrackingRefiner._crop_image / refiner_utils.update_K_with_crop / _get_threshold_for_confidence -> the mirror's host arithmetic   bit for bit
  merge     scripts/merge_results.py (run as a script on a results tree)                                   -> scripts.merge_results                     same files, same rows
Shims for packages the image lacks are gen_golden's (none of them is on the functions under test)."""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

from oracle import fp_oracle as fo
from oracle.gen_golden import REF, install_shims


def main():
    assert REF.exists(), "/root/reference is only present in the build container"
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    # host-side product code under test is imported BEFORE the shims drop the repo's alias packages from the path
    from freepose_amd.src.pipeline.utils import mask_to_rle_pytorch as my_to_rle, rle_to_mask as my_from_rle, z_from_extents
    from freepose_amd.src.utils.bbox_utils import unresizable_box
    from freepose_amd.scripts.dino_inference import CSV_COLUMNS, pose_row
    from freepose_amd.src.pipeline import refiner_utils as my_ru
    from freepose_amd.scripts import merge_results as my_merge
    from freepose_amd.src.pipeline.estimators.tracking_refiner import TrackingRefiner as MyRefiner
    install_shims()
    from src.utils.bbox_utils import CropResizePad
    from src.pipeline.utils import Proposals, depthmap_to_pointcloud, get_z_from_pointcloud, mask_to_bbox
    from sam2.utils.amg import mask_to_rle_pytorch, rle_to_mask
    import src.pipeline.estimators.pose_estimator as pe_mod
    import src.pipeline.estimators.online_pose_estimator as on_mod
    torch.set_num_threads(8)
    print(f"oracle vs reference, {n_cases} random cases per section, seed {seed}")

    def rng_of(section, case):
        return np.random.Generator(np.random.PCG64([seed, sum(map(ord, section)), case]))

    # ---- crop ------------------------------------------------------------------------------------------------------------
    t0, n_box, n_raise = time.time(), 0, 0
    for case in range(n_cases):
        rng = rng_of("crop", case)
        H, W = int(rng.integers(12, 260)), int(rng.integers(12, 260))
        target = int(rng.choice([14, 30, 42, 98, 224]))
        ext = float(rng.choice([0, 0.05, 0.1, 0.2, 0.5]))
        img = rng.random((3, H, W)).astype(np.float32)
        boxes = []
        for _ in range(int(rng.integers(1, 9))):
            kind = rng.integers(0, 4)
            x0, y0 = int(rng.integers(0, W - 4)), int(rng.integers(0, H - 4))
            if kind == 0:
                boxes.append([x0, y0, int(rng.integers(x0 + 2, W + 1)), int(rng.integers(y0 + 2, H + 1))])
            elif kind == 1:
                boxes.append([x0, y0, x0 + int(rng.integers(1, 4)), int(rng.integers(y0 + 1, H + 1))])      # a sliver
            elif kind == 2:
                boxes.append([0, 0, W, H])
            else:
                boxes.append([x0, y0, min(W, x0 + int(rng.integers(3, 60))), min(H, y0 + int(rng.integers(3, 60)))])
        boxes = np.array(boxes, dtype=np.int32)
        proc = CropResizePad(target, (H, W), bbox_extend=ext)
        for b in boxes:                                       # one box at a time: a box the reference cannot resize must fail in the oracle too
            try:
                want = proc(torch.from_numpy(img)[None], torch.from_numpy(b[None])).numpy()
                made = want.shape[-1] == target               # (an exactly square crop can come out one pixel short: unusable downstream)
            except Exception:
                made = False
            if not made:
                n_raise += 1
                assert unresizable_box(b[None], H, W, target, ext) == 0, f"crop case {case}: the mirror's host check lets box {b.tolist()} through"
                try:
                    fo.crop_resize_pad(img[None], b[None], target, ext)
                except ValueError:
                    continue
                raise AssertionError(f"crop case {case}: the reference raises on box {b.tolist()} ({H}x{W}, target {target}, ext {ext}), the oracle does not")
            assert unresizable_box(b[None], H, W, target, ext) == -1, f"crop case {case}: the mirror's host check refuses box {b.tolist()}"
            got = fo.crop_resize_pad(img[None], b[None], target, ext)
            assert np.array_equal(got, want), f"crop case {case}: box {b.tolist()} image {H}x{W} target {target} ext {ext}: {np.abs(got - want).max()}"
            n_box += 1
    print(f"crop      {n_cases} cases, {n_box} boxes byte-identical, {n_raise} boxes the reference cannot crop to the target size: refused by the oracle and the mirror too   ({time.time() - t0:.0f} s)")

    # ---- proposals -----------------------------------------------------------------------------------------------------------
    t0, n_prop, n_short = time.time(), 0, 0
    for case in range(n_cases):
        rng = rng_of("proposals", case)
        Hi, Wi = int(rng.integers(40, 300)), int(rng.integers(40, 300))
        image = rng.integers(0, 256, size=(Hi, Wi, 3), dtype=np.uint8)
        n = int(rng.integers(1, 5))
        yy, xx = np.mgrid[0:Hi, 0:Wi]
        masks, boxes = [], []
        for _ in range(n):
            cy, cx = rng.uniform(0.2, 0.8) * Hi, rng.uniform(0.2, 0.8) * Wi
            ry, rx = rng.uniform(0.08, 0.3) * Hi, rng.uniform(0.08, 0.3) * Wi
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
            if rng.integers(0, 3) == 0:
                m &= rng.random((Hi, Wi)) < 0.8                                         # a mask with holes
            ys, xs = np.nonzero(m)
            masks.append(m)
            boxes.append([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1])
        masks, boxes = np.stack(masks), np.array(boxes, dtype=np.int64)
        res = int(rng.choice([28, 56, 98]))
        ext = float(rng.choice([0.05, 0.1, 0.2]))
        mask_rgb = bool(rng.integers(0, 2))
        try:
            p = Proposals(image, {"masks": torch.from_numpy(masks), "boxes": torch.from_numpy(boxes)}, res, 1, 2, bbox_extend=ext, mask_rgb=mask_rgb)
            sizes_ok = tuple(p.proposals.shape[-2:]) == (res, res)
        except RuntimeError:                                   # torch.stack of crops of unequal size (an exactly square box can come out one pixel short)
            sizes_ok = False
        if not sizes_ok:
            n_short += 1
            assert unresizable_box(boxes, Hi, Wi, res, ext) >= 0, f"proposals case {case}: the mirror's host check lets the detections through"
            try:
                fo.crop_resize_pad(image[None], boxes.astype(np.int32), res, ext, masks.astype(np.uint8), 0, u8_float_div=True)
            except ValueError:
                continue
            raise AssertionError(f"proposals case {case}: the reference cannot make {res} px crops of these detections, the oracle does")
        assert unresizable_box(boxes, Hi, Wi, res, ext) == -1, f"proposals case {case}: the mirror's host check refuses good detections"
        rgb = fo.crop_resize_pad(image[None], boxes.astype(np.int32), res, ext, masks.astype(np.uint8), 1 if mask_rgb else 0, u8_float_div=True)
        mm = fo.crop_resize_pad(image[None], boxes.astype(np.int32), res, ext, masks.astype(np.uint8), 2, u8_float_div=True)
        assert np.array_equal(rgb, p.proposals.numpy()), f"proposals case {case}: crops differ ({Hi}x{Wi}, res {res}, ext {ext}, mask_rgb {mask_rgb})"
        assert np.array_equal(mm[:, 0] > 0.5, p.proposals_masks.numpy()), f"proposals case {case}: crop masks differ"
        n_prop += n
    print(f"proposals {n_cases} cases, {n_prop} proposals: crops and crop masks byte-identical; {n_short} cases where the reference cannot make crops of "
          f"the target size (refused by the oracle and the mirror too)   ({time.time() - t0:.0f} s)")

    # ---- rle -------------------------------------------------------------------------------------------------------------
    t0 = time.time()
    for case in range(n_cases):
        rng = rng_of("rle", case)
        H, W, n = int(rng.integers(1, 120)), int(rng.integers(1, 120)), int(rng.integers(1, 5))
        m = rng.random((n, H, W)) < float(rng.choice([0.0, 0.02, 0.5, 0.98, 1.0]))
        ref = mask_to_rle_pytorch(torch.from_numpy(m))
        mine = my_to_rle(m)
        for a, b, mk in zip(ref, mine, m):
            assert list(a["size"]) == list(b["size"]) and list(a["counts"]) == list(b["counts"]), f"rle case {case}"
            assert np.array_equal(my_from_rle(a), mk) and np.array_equal(rle_to_mask(b), mk), f"rle case {case}: decode"
    print(f"rle       {n_cases} cases: encoder and decoder equal to sam2's, both directions   ({time.time() - t0:.0f} s)")

    # ---- depth -> point cloud -> pose ------------------------------------------------------------------------------------------
    t0, worst_ext, worst_t = time.time(), 0.0, 0.0
    grid = np.array(pe_mod.DinoPoseEstimator.generate_poses(64))
    for case in range(n_cases):
        rng = rng_of("depth", case)
        depth = np.zeros((420, 420), np.float32)
        y0, x0 = int(rng.integers(0, 400)), int(rng.integers(0, 400))
        h, w = int(rng.integers(1, 420 - y0 + 1)), int(rng.integers(1, 420 - x0 + 1))
        if rng.integers(0, 6) == 0:
            h, w = min(h, 9), min(w, 9)                                                  # fewer than 100 px: the fallback square
        patch = (0.3 + 2.0 * rng.random((h, w))).astype(np.float32)
        patch[rng.random((h, w)) < float(rng.choice([0.0, 0.3, 0.9]))] = 0
        depth[y0:y0 + h, x0:x0 + w] = patch
        if not (depth > 0).any():
            depth[y0, x0] = 1.0
        K420 = np.array([[600, 0, 210], [0, 600, 210], [0, 0, 1]], dtype=np.int64)
        ext = fo.depth_extents(depth[None], 600, 600, 210, 210)[0]
        pc = depthmap_to_pointcloud(depth, K420)
        assert len(pc) == int(ext[6]), f"depth case {case}: point count"
        ex, ey = pc[:, 0].max() - pc[:, 0].min(), pc[:, 1].max() - pc[:, 1].min()
        worst_ext = max(worst_ext, abs(ext[4] - ex), abs(ext[5] - ey))
        mask = depth > 0
        if mask.sum() < 100:
            mask[105:315, 105:315] = True                                                # template.py:75-77, renderer.py:116-117
        bb = mask_to_bbox(mask)
        assert list(ext[:4]) == list(bb), f"depth case {case}: mask bbox {list(ext[:4])} vs {list(bb)}"
        Kq = np.array([[rng.uniform(400, 1200), 0, rng.uniform(200, 400)], [0, rng.uniform(400, 1200), rng.uniform(150, 300)], [0, 0, 1]])
        bx0, by0 = rng.uniform(0, 400), rng.uniform(0, 300)
        bbox = np.array([bx0, by0, bx0 + rng.uniform(5, 300), by0 + rng.uniform(5, 300)])
        est = float(rng.uniform(0.02, 0.5))
        pose = grid[int(rng.integers(0, 64))]
        mean = pc.mean(axis=0)
        pc2 = (pc - mean) / 0.25 * est + mean
        want = get_z_from_pointcloud(bbox, pc2, Kq, pose)
        got = z_from_extents(bbox, ext[4] * est / 0.25, ext[5] * est / 0.25, Kq, pose)
        worst_t = max(worst_t, float(np.abs(np.asarray(got) - np.asarray(want)).max()))
    assert worst_ext < 1e-12 and worst_t < 1e-9, (worst_ext, worst_t)
    print(f"depth     {n_cases} cases: point counts and mask boxes equal, extents within {worst_ext:.1e}, poses within {worst_t:.1e} m   ({time.time() - t0:.0f} s)")

    # ---- rotation grids + geodesic neighbourhoods ----------------------------------------------------------------------------------
    t0, worst_R, n_sel = time.time(), 0.0, 0
    for case in range(max(1, n_cases // 10)):
        rng = rng_of("poses", case)
        n = int(rng.choice([1, 2, 3, int(rng.integers(4, 3000)), int(rng.integers(3000, 30000))]))
        ref = np.array(pe_mod.DinoPoseEstimator.generate_poses(n))
        R = fo.generate_rotations(n)
        worst_R = max(worst_R, float(np.abs(R - ref[:, :3, :3]).max()))
        assert (ref[:, 3] == [0, 0, 0, 1]).all()
        for _ in range(10):
            q = np.eye(4)
            if rng.integers(0, 2):
                q[:3, :3] = ref[int(rng.integers(0, n)), :3, :3]
            else:
                a, _r = np.linalg.qr(rng.standard_normal((3, 3)))
                q[:3, :3] = a * np.sign(np.linalg.det(a))
            thr = float(rng.choice([rng.uniform(1, 40), rng.uniform(40, 180)]))
            d = on_mod.DinoOnlinePoseEstimator.geodesic_distance(ref[:, :3, :3], q)
            if np.abs(d - thr).min() < 1e-9:
                continue                                                                 # (a distance ON the threshold: not decidable in floating point)
            want = np.where(d < thr)[0]
            assert np.array_equal(fo.geodesic_select(ref[:, :3, :3], q, thr), want), f"geodesic case {case}: n {n} thr {thr}"
            n_sel += 1
    assert worst_R <= 1e-15, worst_R
    print(f"poses     {max(1, n_cases // 10)} grids within {worst_R:.1e} of generate_poses; geodesic: {n_sel} neighbourhoods, identical index lists   ({time.time() - t0:.0f} s)")
    # ---- CSV rows of the two pose drivers ------------------------------------------------------------------------------------------
    import io
    import types
    import pandas as pd
    from scipy.spatial.transform import Rotation as Rot
    from oracle.gen_golden_r2 import _ref_block
    t0 = time.time()
    img_block = _ref_block(REF / "scripts" / "dino_inference.py", 'results_dict["scene_id"].append(int(scene_id))', 'results_dict["time"].append(0.2)')
    vid_block = _ref_block(REF / "scripts" / "dino_inference_video.py", 'results_dict["scene_id"].append(0)', 'results_dict["time"].append(-1)')
    keys = ["scene_id", "im_id", "obj_id", "score", "R", "t", "bbox_visib", "scale", "time"]
    for case in range(n_cases):
        rng = rng_of("csv", case)
        T = np.eye(4)
        T[:3, :3] = Rot.from_rotvec(rng.standard_normal(3) * float(rng.choice([1e-4, 1.0, 3.0]))).as_matrix()
        T[:3, 3] = rng.standard_normal(3) * float(rng.choice([1e-6, 0.1, 5.0, 1e4]))
        if rng.integers(0, 8) == 0:
            T[:3, 3] = [0.0, -0.0, 1.0]
        score = np.float32(torch.tensor(float(rng.uniform(-1, 1))).to(torch.bfloat16).float().item())
        out = {"TCO": [T, np.eye(4), np.eye(4)], "scores": np.array([score, 0.5, 0.25], dtype=np.float32),
               "bbox": torch.tensor([int(v) for v in rng.integers(0, 2000, 4)])}
        mesh = str(rng.choice(["0a1b2c3d4e5f60718293a4b5c6d7e8f9", "Shark", "mug_01", "7"]))
        scale = float(rng.choice([0.05, 0.1234567, 1e-3, 0.3, 1 / 3]))
        scene, frame = str(int(rng.integers(0, 100))), int(rng.integers(0, 5000))
        rd = {k: [] for k in keys}
        exec(img_block, {"np": np}, {"results_dict": rd, "scene_id": scene, "frame_id": frame, "proposals": types.SimpleNamespace(meshes=[mesh]),
                                     "prop_idx": 0, "out": out, "scales": [scale]})
        want, got = io.StringIO(), io.StringIO()
        pd.DataFrame(rd).to_csv(want, index=False, header=True)
        pd.DataFrame([pose_row(scene, frame, mesh, out["scores"][0], T, out["bbox"].numpy(), scale)], columns=CSV_COLUMNS).to_csv(got, index=False, header=True)
        assert got.getvalue() == want.getvalue(), f"csv case {case} (image driver):\n{got.getvalue()}\n{want.getvalue()}"
        rd = {k: [] for k in keys}
        exec(vid_block, {"np": np}, {"results_dict": rd, "frame_idx": frame, "mesh_id": mesh, "out": out, "scales": [scale], "obj_idx": 0})
        want, got = io.StringIO(), io.StringIO()
        pd.DataFrame(rd).to_csv(want, index=False, header=True)
        pd.DataFrame([pose_row(0, frame, mesh, out["scores"][0], T, out["bbox"].numpy(), scale, t_scale=1, time_value=-1)],
                     columns=CSV_COLUMNS).to_csv(got, index=False, header=True)
        assert got.getvalue() == want.getvalue(), f"csv case {case} (video driver):\n{got.getvalue()}\n{want.getvalue()}"
    print(f"csv       {n_cases} cases: rows of both drivers character for character   ({time.time() - t0:.0f} s)")

    # ---- bf16 score expressions (the reference leaves the reduction order to torch: agreement is statistical, never worse than 1 ulp) ----
    import torch.nn.functional as F
    from einops import einsum
    t0, same, total, worst = time.time(), 0, 0, 0.0
    decisive, agree = 0, 0
    for case in range(max(1, n_cases // 20)):
        rng = rng_of("score", case)
        Tn, P, D = int(rng.integers(2, 40)), int(rng.choice([36, 256, 900])), int(rng.choice([64, 384, 1024]))
        base = rng.standard_normal((P, D)).astype(np.float32)
        tm = torch.from_numpy((base[None] * 0.5 + rng.standard_normal((Tn, P, D))).astype(np.float32)).to(torch.bfloat16)
        q = torch.from_numpy((base + 0.7 * rng.standard_normal((P, D))).astype(np.float32)).to(torch.bfloat16)
        if rng.integers(0, 2):                                                            # a planted answer: one template is a noisy copy of the query
            tm[int(rng.integers(0, Tn))] = (q.float() + 0.5 * torch.from_numpy(rng.standard_normal((P, D)).astype(np.float32))).to(torch.bfloat16)
        ref = einsum(F.normalize(tm, dim=-1), F.normalize(q[None], dim=-1).expand(Tn, -1, -1), "b n d, b n d -> b n").mean(dim=-1).float().numpy()   # pose_estimator.py:85-88
        s = fo.template_score(fo.torch_to_bits(tm), fo.l2norm_rows(fo.torch_to_bits(q)))
        ulp = np.abs(ref) * 2.0 ** -7 + 1e-9
        worst = max(worst, float((np.abs(s - ref) / ulp).max()))
        same += int((s == ref).sum())
        total += Tn
        top2 = np.sort(ref)[::-1][:2]
        if top2[0] - top2[1] > 2 * ulp.max():
            decisive += 1
            agree += int(s.argmax() == ref.argmax())
    assert worst <= 1.0 and agree == decisive, (worst, agree, decisive)
    print(f"score     {max(1, n_cases // 20)} template sets: {same}/{total} scores bit-identical to the reference's torch expression, all within "
          f"{worst:.2f} bf16 ulp; arg-max identical on all {decisive} decisive sets   ({time.time() - t0:.0f} s)")

    t0, same, total, n_set = time.time(), 0, 0, 0
    for case in range(max(1, n_cases // 50)):
        rng = rng_of("retrieval", case)
        N, D = int(rng.integers(200, 4000)), int(rng.choice([384, 1024]))
        bank = rng.standard_normal((N, D)).astype(np.float32) + 2.0 * rng.standard_normal(D).astype(np.float32)
        rf = F.normalize(torch.from_numpy(bank).to(torch.bfloat16), dim=-1)                   # extract_proposals_ground.py:40-41
        feat = torch.from_numpy(rng.standard_normal((900, D)).astype(np.float32)).to(torch.bfloat16)
        m = rng.random(900) < rng.uniform(0.05, 0.9)
        m[0] = True
        qref = F.normalize(feat[torch.from_numpy(m)].mean(dim=0)[None], dim=-1)[0]            # :130-134
        ffa_bits, _ = fo.ffa(fo.torch_to_bits(feat)[None], m.astype(np.uint8).reshape(1, 1, 900), 1)
        d = np.abs(fo.from_bf16_bits(ffa_bits[0]) - feat[torch.from_numpy(m)].mean(dim=0).float().numpy())
        assert (d <= np.abs(fo.from_bf16_bits(ffa_bits[0])) * 2.0 ** -7 + 1e-6).all(), f"retrieval case {case}: FFA"
        ref = (rf @ qref).float().numpy()                                                     # :137
        bank_bits = fo.bank_prepare(bank)
        sc = fo.bank_scores(bank_bits, fo.torch_to_bits(qref[None])[0])
        same += int((sc == ref).sum())
        total += N
        assert (np.abs(sc - ref) <= np.abs(ref) * 2.0 ** -5 + 1e-6).all(), f"retrieval case {case}: scores"
        k = min(100, N)
        s, i = fo.bank_topk(bank_bits, fo.torch_to_bits(qref[None]), k)
        top = torch.topk(torch.from_numpy(ref), k)                                            # :140
        assert np.allclose(np.sort(s[0]), np.sort(top.values.numpy()), atol=2.0 ** -8), f"retrieval case {case}: top-k scores"
        thr = top.values.numpy().min()
        assert (ref[i[0]] >= thr - abs(thr) * 2.0 ** -7).all(), f"retrieval case {case}: top-k members"
        n_set += 1
    assert same / total >= 0.998
    print(f"retrieval {n_set} banks: {same}/{total} bank scores bit-identical to torch's bf16 matmul (the rest: rows whose bf16 norm flips under torch's "
          f"reduction order), top-100 = the same score multiset, every member above the reference's 100th score   ({time.time() - t0:.0f} s)")
    # ---- TrackingRefiner host arithmetic (crop boxes, RoIs, cropped intrinsics, confidence threshold) ---------------------------------
    t0 = time.time()
    tv = sys.modules["torchvision"]

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
    tv.transforms.ToTensor, tv.transforms.Compose = ToTensor, Compose
    recorded = {}

    def roi_align(image, boxes, output_size, sampling_ratio=-1, **kw):           # torchvision's operator is not the reference's code: record its arguments
        recorded["rois"] = boxes.clone()
        return torch.zeros((boxes.shape[0], image.shape[1]) + tuple(output_size))
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.roi_align = roi_align
    sys.modules["torchvision.ops"] = tv.ops
    sys.modules["open3d"] = types.ModuleType("open3d")
    sys.modules["cv2"].INTER_CUBIC = 2
    from src.pipeline import refiner_utils
    from src.pipeline.estimators.tracking_refiner import TrackingRefiner
    ref_obj, my_obj = TrackingRefiner.__new__(TrackingRefiner), MyRefiner.__new__(MyRefiner)
    n_ref = max(1, n_cases // 10)
    for case in range(n_ref):
        rng = rng_of("refiner", case)
        verts = (rng.standard_normal((int(rng.integers(100, 4000)), 3)) * rng.uniform(0.01, 0.3, 3)).astype(np.float64)
        mesh = types.SimpleNamespace(vertices=verts)
        K = np.array([[rng.uniform(400, 1400), 0, rng.uniform(200, 500)], [0, rng.uniform(400, 1400), rng.uniform(150, 400)], [0, 0, 1]])
        T = np.eye(4)
        T[:3, :3] = Rot.from_rotvec(rng.standard_normal(3) * 2).as_matrix()
        T[:3, 3] = rng.uniform([-0.4, -0.3, 0.3], [0.4, 0.3, 3.0])
        Hh, Ww = int(rng.integers(200, 800)), int(rng.integers(200, 1000))
        image = torch.zeros((3, Hh, Ww))
        crop, bbox, newK = ref_obj._crop_image(mesh, image, K, T)
        pts = MyRefiner._sample_points(mesh)
        Kt = torch.from_numpy(K).view(3, 3).float()
        boxes = my_ru.crop_boxes(torch.from_numpy(T).view(1, 4, 4).float(), pts, Kt, 518, 518)
        assert np.array_equal(boxes.numpy()[0], bbox.numpy()), f"refiner case {case}: crop box"
        assert np.array_equal(torch.cat([torch.zeros((1, 1)), boxes], 1).numpy(), recorded["rois"].numpy()), f"refiner case {case}: RoI"
        assert np.array_equal(my_ru.update_K_with_crop(Kt, boxes, 518, 518).numpy()[0], newK.numpy()), f"refiner case {case}: intrinsics"
        sims = rng.random((int(rng.integers(1, 9)), 37, 37)).astype(np.float32) * (rng.random((1, 37, 37)) > rng.uniform(0, 0.9))
        for qq in (0.2, 0.05, 0.5):
            assert float(my_obj._get_threshold_for_confidence(sims, top_quantile=qq)) == float(ref_obj._get_threshold_for_confidence(sims, top_quantile=qq)), \
                f"refiner case {case}: threshold {qq}"
    print(f"refiner   {n_ref} cases: sampled points -> crop box, RoI, cropped intrinsics and confidence thresholds equal to the reference's, bit for bit   ({time.time() - t0:.0f} s)")
    # ---- merge_results: the reference's script run on a results tree, beside the mirror ------------------------------------------------
    import runpy
    import tempfile
    t0 = time.time()
    n_merge = max(1, n_cases // 100)
    for case in range(n_merge):
        rng = rng_of("merge", case)
        with tempfile.TemporaryDirectory() as td:
            td = Path(td)
            ds, split = str(rng.choice(["ycbv", "tless", "lmo"])), "test"
            for fi in range(int(rng.integers(1, 4))):
                folder = td / "data" / "results" / ds / f"props-{fi}_{ds}-{split}_dinopose_layer_22_bbext_0.05"
                folder.mkdir(parents=True)
                for task in range(int(rng.integers(1, 6))):
                    n_rows = int(rng.integers(0, 5))
                    df = pd.DataFrame([{"scene_id": 1, "im_id": 100 * task + r, "obj_id": "m", "score": float(rng.random()), "R": "1 0 0 0 1 0 0 0 1",
                                        "t": "0 0 1.5", "bbox_visib": "1 2 3 4", "scale": 0.1, "time": 0.2} for r in range(n_rows)], columns=CSV_COLUMNS)
                    if n_rows and rng.integers(0, 4) == 0:
                        df.loc[0, "t"] = None
                    df.to_csv(folder / f"pose_outputs_{task}.csv", index=False)
            (td / "data" / "results" / ds / "props.json").write_text("[]")
            cwd, argv = os.getcwd(), sys.argv
            out_ref, out_my = td / "ref", td / "mine"
            out_ref.mkdir()
            out_my.mkdir()
            (out_ref / "data").symlink_to(td / "data")
            (out_my / "data").symlink_to(td / "data")
            try:
                os.chdir(out_ref)
                sys.argv = ["merge_results.py", "--dataset", ds]
                try:
                    runpy.run_path(str(REF / "scripts" / "merge_results.py"), run_name="__main__")
                    ref_ok = True
                except ValueError:                              # pd.concat of nothing: a folder without a single row
                    ref_ok = False
                os.chdir(out_my)
                try:
                    my_merge.main(["--dataset", ds])
                    my_ok = True
                except ValueError:
                    my_ok = False
            finally:
                os.chdir(cwd)
                sys.argv = argv
            assert ref_ok == my_ok, f"merge case {case}: failure behaviour differs"
            if ref_ok:
                a = sorted(p.name for p in out_ref.glob("*.csv"))
                b = sorted(p.name for p in out_my.glob("*.csv"))
                assert a == b and a, f"merge case {case}: file names {a} vs {b}"
                for name in a:
                    ra = sorted((out_ref / name).read_text().splitlines()[1:])
                    rb = sorted((out_my / name).read_text().splitlines()[1:])
                    assert ra == rb and (out_ref / name).read_text().splitlines()[0] == (out_my / name).read_text().splitlines()[0], f"merge case {case}: rows of {name}"
    print(f"merge     {n_merge} result trees: the reference's scripts/merge_results.py and the mirror write the same files with the same rows (the mirror in (task, rank) "
          f"order, the reference in directory order)   ({time.time() - t0:.0f} s)")
    print("all sections green")


if __name__ == "__main__":
    main()
