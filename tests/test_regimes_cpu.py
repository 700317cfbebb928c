"""The video script's precision regime against the static-image regime, on the oracle (CPU only).

The reference runs the SAME estimator code under two precision regimes: `scripts/dino_inference.py` casts the model
to bf16 and feeds bf16 crops; `scripts/dino_inference_video.py:151` additionally wraps the calls in `torch.autocast(bf16)` and
`DinoOnlinePoseEstimator.forward_fine` feeds **fp16** crops (online_pose_estimator.py:51,66).  freepose_amd runs the video path in
the static regime (bf16 crops, bf16 features: DESIGN §4 "deviations kept on purpose").  This test puts a number on that deviation
with nothing but the oracle: a render-and-compare scene (textured box, 48 hypotheses, 6 perturbed queries) scored in both regimes.

Stated tolerance: every score of the video regime is within 4 bf16 ulps of the static regime's score for the same hypothesis, and
wherever the static regime's best hypothesis leads its runner-up by more than 4 ulps the two regimes pick the same hypothesis (the
same discrete rotation, hence identical R and — extents being regime-independent — identical t)."""
import numpy as np
import torch

from tests._meshes import checker_gradient_texture, textured_cube

N_HYP, N_QUERY, RES, LAYER = 48, 6, 224, 22
TOL_ULP = 4


def _ulp_bf16(x):
    x = np.abs(np.asarray(x, dtype=np.float32))
    return (2.0 ** (np.floor(np.log2(np.maximum(x, 2.0 ** -126))) - 7)).astype(np.float32)


def test_video_regime_agrees_with_static_regime_on_the_oracle(capsys):
    from freepose_amd import ops
    from freepose_amd.src.pipeline.retrieval.renderer import grid_poses
    from oracle import fp_oracle as fo, vit_ref
    from scipy.spatial.transform import Rotation as Rot

    v, f, uv = textured_cube()
    v = v * np.array([1.0, 0.7, 0.45], np.float32)
    tex = checker_gradient_texture(256)
    sd = ops.random_state_dict("dinov2_vits14_reg", seed=3)
    poses = np.array(grid_poses(N_HYP))
    fx = fy = 600.0
    cx = cy = 210.0
    rng = np.random.Generator(np.random.PCG64(17))
    picks = rng.choice(N_HYP, size=N_QUERY, replace=False)
    q_poses = []
    for j in picks:
        P = np.array(poses[j], dtype=np.float64)
        ax = rng.standard_normal(3)
        P[:3, :3] = Rot.from_rotvec(np.deg2rad(rng.uniform(3.0, 7.0)) * ax / np.linalg.norm(ax)).as_matrix() @ P[:3, :3]
        P[:3, 3] += [rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03), rng.uniform(-0.08, 0.08)]
        q_poses.append(P)
    allp = np.concatenate([poses, np.array(q_poses)]).astype(np.float32)
    rgb, depth = fo.rasterize(v, f, None, allp, 0.25, fx, fy, cx, cy, 420, 420, uv=uv, texture=tex)
    ext = fo.depth_extents(depth, fx, fy, cx, cy)
    crops = torch.from_numpy(fo.crop_resize_pad(rgb, ext[:, :4].astype(np.int32), RES, 0.0))          # f32 [H+Q,3,RES,RES], values k/255

    with torch.inference_mode():
        # static regime (dino_inference.py:46, pose_estimator.py:21,84-90): bf16 model, bf16 crops, bf16 features and rounding points
        fa = vit_ref.vit_forward(sd, crops.to(torch.bfloat16).float(), layer=LAYER, feature_type="patch", dtype=torch.bfloat16).to(torch.bfloat16)
        # video regime (online_pose_estimator.py:51,66 + dino_inference_video.py:151): fp16 crops, autocast, fp32 features
        fb = vit_ref.vit_forward_video_regime(sd, crops, layer=LAYER, feature_type="patch")
    fa_bits = fo.torch_to_bits(fa)
    agree = decisive = 0
    worst = 0.0
    for b in range(N_QUERY):
        s_a = fo.template_score(fa_bits[:N_HYP], fo.l2norm_rows(fa_bits[N_HYP + b]))
        s_b = vit_ref.score_video_regime(fb[N_HYP + b], fb[:N_HYP]).numpy()
        d = float(np.max(np.abs(s_a - s_b) / _ulp_bf16(s_a)))
        worst = max(worst, d)
        order = np.lexsort((np.arange(N_HYP), -s_a))
        lead = float((s_a[order[0]] - s_a[order[1]]) / _ulp_bf16(s_a[order[0]]))
        top_b = int(np.lexsort((np.arange(N_HYP), -s_b))[0])
        if lead > TOL_ULP:
            decisive += 1
            assert top_b == int(order[0]), f"query {b}: regimes disagree on a decisive query (lead {lead:.1f} ulp)"
        else:
            assert top_b in [int(i) for i in order[:3]], f"query {b}: video-regime pick outside the static regime's top-3"
        agree += int(top_b == int(order[0]))
    with capsys.disabled():
        print(f"\n[regimes] video (fp16 in, autocast) vs static (bf16): top-1 agreement {agree}/{N_QUERY}, decisive {decisive}/{N_QUERY}, "
              f"worst |score difference| {worst:.2f} bf16 ulp")
    assert worst <= TOL_ULP, worst
    assert decisive >= N_QUERY // 2, "scene too ambiguous to say anything"
