"""End-to-end pose parity with the ViT in the loop (VERDICT r2 "missing" #1, north star: "output R,t poses match the reference
within a stated tolerance on identical proposals ... within 2 deg / 2 mm").

BASELINE config 4: ViT-L/14-reg layer 22 (seeded random-init weights of the real architecture), pose hypotheses of a TEXTURED mesh,
queries = renders of the same mesh at perturbed hypothesis poses.  Two sizes, both in the default `-m gpu` run:
  * reduced — 64 hypotheses, 8 queries, 420^2 crops, oracle ViT in bf16 (the reference's regime) AND fp32   (this file);
  * the configuration's OWN size — 576 hypotheses, 518^2 crops, 2 queries, bf16 regime   (tests/test_gpu_zz_pose_parity_full.py, which
    sorts last so its table ends the suite's output; FP_PARITY_QUERIES / FP_PARITY_FP32=1 widen it).

  oracle side (CPU, nothing from freepose_amd):  fo.rasterize -> fo.depth_extents -> fo.crop_resize_pad -> vit_ref.vit_forward
      (fp32, and the reference's bf16 regime: bf16 weights/activations, pose_estimator.py:21) -> fo.template_score
      (pose_estimator.py:85-90 with its bf16 rounding points) -> top-3 (score desc, index asc) -> z_from_extents
      (pose_estimator.py:104-116, get_z_from_pointcloud)
  HIP side:  freepose_amd.pipeline.HotPath on the SAME query crops / boxes / intrinsics.

Everything except the ViT is bit-exact between the two sides (asserted here again on the hypothesis crops and extents), so the
only source of disagreement is feature noise (bf16 storage / summation order), which moves each score by a few bf16 ulps.

Stated tolerance:
  * a query is DECISIVE when the oracle's best score leads its runner-up by more than MARGIN_ULP bf16 ulps; on decisive queries the
    HIP arg-max must equal the oracle's, hence R is identical (re = 0) and t agrees to 1e-6 m (bit-exact extents);
  * on non-decisive queries (a tie within feature noise: torch.argmax's own tie order is unspecified there) the
    HIP pick must be one of the oracle's top-3 and its oracle score within MARGIN_ULP ulps of the oracle's best;
  * every HIP score is within SCORE_ULP bf16 ulps of the oracle's score for the same hypothesis.
re / te are restated from bop_toolkit_lib/pose_error.py:288-315 and printed for every query together with the agreement rate."""
import math
import os

import numpy as np
import pytest
import torch

from tests._meshes import checker_gradient_texture, textured_cube

pytestmark = pytest.mark.gpu

LAYER = 22
MARGIN_ULP = 3          # bf16 ulps of lead that make an arg-max decisive (scores ~0.3-0.9: 1 ulp = 2^-9 .. 2^-8)
SCORE_ULP = 3           # per-hypothesis |HIP score - oracle score| bound, bf16 ulps of the oracle score


def re_deg(R_est, R_gt):
    """bop_toolkit_lib/pose_error.py:288-303"""
    c = float(0.5 * (np.trace(R_est.dot(np.linalg.inv(R_gt))) - 1.0))
    return 180.0 * math.acos(min(1.0, max(-1.0, c))) / np.pi


def te_m(t_est, t_gt):
    """bop_toolkit_lib/pose_error.py:306-315"""
    return float(np.linalg.norm(np.asarray(t_gt).reshape(3) - np.asarray(t_est).reshape(3)))


def _ulp_bf16(x):
    x = np.abs(np.asarray(x, dtype=np.float32))
    e = np.floor(np.log2(np.maximum(x, 2.0 ** -126)))
    return (2.0 ** (e - 7)).astype(np.float32)


def _rot(axis, deg):
    from scipy.spatial.transform import Rotation as Rot
    axis = np.asarray(axis, dtype=np.float64)
    return Rot.from_rotvec(np.deg2rad(deg) * axis / np.linalg.norm(axis)).as_matrix()


def _oracle_feats(sd, crops_f32, dtype, batch=8):
    from oracle import vit_ref
    out = []
    nthr = torch.get_num_threads()
    # 16 threads: the fastest count on the pool's hosts (bench.py's cpu_baseline probe).  More threads are slower, and so are several
    # pinned 16-thread worker processes (measured in round 6: 0.38 s per crop with one worker, 0.53 s per crop and worker with eight —
    # the box's CPU quota, not its 256 logical CPUs, sets the rate), so the oracle runs in this process.
    torch.set_num_threads(min(16, nthr))
    try:
        with torch.inference_mode():
            for i in range(0, crops_f32.shape[0], batch):
                out.append(vit_ref.vit_forward(sd, crops_f32[i:i + batch], layer=LAYER, feature_type="patch", dtype=dtype).to(torch.bfloat16))
    finally:
        torch.set_num_threads(nthr)
    return torch.cat(out)


def test_pose_parity_vit_in_the_loop(capsys):
    """reduced size, both regimes"""
    run_pose_parity(capsys, 64, 8, 420, ("bf16", "fp32"))


def run_pose_parity(capsys, N_HYP, N_QUERY, RES, regime_names):
    from freepose_amd import ops
    from freepose_amd.pipeline import HotPath
    from freepose_amd.retrieval import TemplateBank
    from freepose_amd.src.pipeline.utils import z_from_extents
    from oracle import fp_oracle as fo
    import bench

    v, f, uv = textured_cube()
    v = v * np.array([1.0, 0.7, 0.45], np.float32)             # a box with three different extents: poses are distinguishable
    tex = checker_gradient_texture(256)
    ck = os.environ.get("FP_PARITY_STATE_DICT")        # tools/real_weights_parity.py: the same test on the trained checkpoint
    if ck:
        sd = torch.load(ck, map_location="cpu")
        sd = sd["model"] if isinstance(sd, dict) and "model" in sd and "pos_embed" not in sd else sd
        sd = {k: v.to(torch.bfloat16) for k, v in sd.items() if k != "mask_token"}
    else:
        sd = ops.random_state_dict("dinov2_vitl14_reg", seed=3)
    vit = ops.ViT("dinov2_vitl14_reg", sd)
    bank = TemplateBank(bench.synthetic_bank(200, 1024, seed=5))
    hp = HotPath(vit, bank, ops.Mesh(v, f, uv=uv, texture=tex), n_hyp=N_HYP, crop_res=RES, render_res=420, k=10, layer=LAYER, vit_batch=32)
    fx, fy, cx, cy = hp.fx, hp.fy, hp.cx, hp.cy
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)

    # ---- queries: the mesh at perturbed hypothesis poses, rendered and cropped by the ORACLE (identical inputs for both sides)
    rng = np.random.Generator(np.random.PCG64(17))
    picks = rng.choice(N_HYP, size=N_QUERY, replace=False)
    q_poses = []
    for j in picks:
        P = np.array(hp.hyp_poses[j], dtype=np.float64)
        P[:3, :3] = _rot(rng.standard_normal(3), rng.uniform(3.0, 7.0)) @ P[:3, :3]
        P[:3, 3] += [rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03), rng.uniform(-0.08, 0.08)]
        q_poses.append(P)
    q_poses = np.array(q_poses)
    q_rgb, q_depth = fo.rasterize(v, f, None, q_poses.astype(np.float32), 0.25, fx, fy, cx, cy, 420, 420, uv=uv, texture=tex)
    q_ext = fo.depth_extents(q_depth, fx, fy, cx, cy)
    q_boxes = q_ext[:, :4].astype(np.int32)
    q_crops = fo.crop_resize_pad(q_rgb, q_boxes, RES, 0.0)                       # f32 [Q,3,420,420]
    q_crops_bf = torch.from_numpy(q_crops).to(torch.bfloat16)
    q_masks = torch.from_numpy(fo.crop_resize_pad(q_rgb, q_boxes, RES, 0.0).sum(1) > 0)
    scales = np.full(N_QUERY, 0.25)

    # ---- HIP side -----------------------------------------------------------------------------------------------------------
    res = hp.run(q_crops_bf.cuda(), q_masks.cuda(), K, q_boxes, scales)
    hyp_crops_g, ext_g = hp.render_hypotheses()
    hyp_feats_g = hp.hypothesis_features(hyp_crops_g)
    q_feats_g = vit(q_crops_bf.cuda(), layer=LAYER, feature_type="patch")
    scores_g = np.stack([ops.template_score(hyp_feats_g, ops.l2_normalize(q_feats_g[b])).cpu().numpy() for b in range(N_QUERY)])
    torch.cuda.synchronize()

    # ---- oracle side --------------------------------------------------------------------------------------------------------
    h_rgb, h_depth = fo.rasterize(v, f, None, hp.hyp_poses.astype(np.float32), 0.25, fx, fy, cx, cy, 420, 420, uv=uv, texture=tex)
    h_ext = fo.depth_extents(h_depth, fx, fy, cx, cy)
    h_crops = fo.crop_resize_pad(h_rgb, h_ext[:, :4].astype(np.int32), RES, 0.0)
    # the non-ViT stages are bit-exact: the hypothesis crops the two ViTs see are the same bits
    assert np.array_equal(fo.torch_to_bits(hyp_crops_g.cpu()), fo.to_bf16_bits(h_crops))
    assert np.array_equal(ext_g.cpu().numpy(), h_ext)
    h_crops_t = torch.from_numpy(h_crops).to(torch.bfloat16).float()             # what the reference's `.to(bf16)` model input holds
    q_crops_t = q_crops_bf.float()
    sd32 = {k: t.float() for k, t in sd.items()}

    report = {}
    regimes = [(n, {"bf16": torch.bfloat16, "fp32": torch.float32}[n]) for n in regime_names]
    for regime, dtype in regimes:
        src = sd if dtype == torch.bfloat16 else sd32
        hf = fo.torch_to_bits(_oracle_feats(src, h_crops_t, dtype))
        qf = fo.torch_to_bits(_oracle_feats(src, q_crops_t, dtype))
        agree = decisive = 0
        worst_ulp, rows = 0.0, []
        for b in range(N_QUERY):
            s_o = fo.template_score(hf, fo.l2norm_rows(qf[b]))
            so_k, io_k = fo.topk_merge(s_o[None], np.arange(N_HYP, dtype=np.int32)[None], 3)
            so_k, io_k = so_k[0], io_k[0]
            lead_ulp = float((so_k[0] - so_k[1]) / _ulp_bf16(so_k[0]))
            d_ulp = float(np.max(np.abs(scores_g[b] - s_o) / _ulp_bf16(s_o)))
            worst_ulp = max(worst_ulp, d_ulp)
            g_top = int(res[b].hyp_idx[0])
            # poses: oracle's top-1 through the reference's formula, HIP's from HotPath
            ratio = float(scales[b]) / 0.25
            T_o = z_from_extents(q_boxes[b], h_ext[io_k[0], 4] * ratio, h_ext[io_k[0], 5] * ratio, K, hp.hyp_poses[io_k[0]])
            T_g = res[b].TCO[0]
            r_err, t_err = re_deg(T_g[:3, :3], T_o[:3, :3]), te_m(T_g[:3, 3], T_o[:3, 3])
            same = g_top == int(io_k[0])
            agree += same
            if lead_ulp > MARGIN_ULP:
                decisive += 1
                assert same, f"[{regime}] query {b}: decisive lead of {lead_ulp:.1f} ulp but HIP picked {g_top}, oracle {io_k[0]}"
            if same:
                assert r_err <= 1e-4 and t_err <= 1e-6, (regime, b, r_err, t_err)   # acos of a trace that is 3 to rounding
            else:
                assert g_top in io_k.tolist(), f"[{regime}] query {b}: HIP pick {g_top} not in the oracle's top-3 {io_k}"
                assert (so_k[0] - s_o[g_top]) / _ulp_bf16(so_k[0]) <= MARGIN_ULP
            # the planted pose is recovered by both (the hypothesis the query was perturbed from is the best or a neighbour)
            rows.append((b, int(picks[b]), int(io_k[0]), g_top, lead_ulp, d_ulp, r_err, t_err * 1e3,
                         re_deg(T_g[:3, :3], q_poses[b][:3, :3]), te_m(T_g[:3, 3], q_poses[b][:3, 3]) * 1e3))
        assert worst_ulp <= SCORE_ULP, f"[{regime}] a HIP score is {worst_ulp:.1f} bf16 ulps from the oracle's"
        report[regime] = (agree, decisive, worst_ulp, rows)

    with capsys.disabled():
        for regime, (agree, decisive, worst_ulp, rows) in report.items():
            print(f"\n[pose parity, {N_HYP} hypotheses @{RES}^2, oracle ViT in {regime}] top-1 agreement {agree}/{N_QUERY} ({decisive} decisive at > {MARGIN_ULP} ulp), "
                  f"worst score difference {worst_ulp:.2f} bf16 ulp")
            print("  query planted oracle hip  lead[ulp] dscore[ulp]  re[deg] te[mm] (HIP vs oracle) | re[deg] te[mm] (HIP vs drawn pose)")
            for r in rows:
                print("  %5d %7d %6d %3d  %9.1f %11.2f  %7.3f %6.3f                  | %7.2f %6.2f" % r)
    # with 3-7 deg perturbations of grid poses most queries must be decisive, or the test shows nothing
    assert all(r[1] >= (N_QUERY + 1) // 2 for r in report.values())
