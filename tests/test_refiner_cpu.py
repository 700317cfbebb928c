"""SURVEY §8(f)-3 host pieces and oracle restatements that need no GPU: RoIAlign (torchvision semantics), OpenCV bicubic
resize, the crop / intrinsics arithmetic of refiner_utils and the confidence threshold.  torchvision and cv2 are not
installable here, so these two restatements are held by known answers (parity unpinned)."""
import numpy as np
import torch

from oracle import fp_oracle as fo


def _ramp(H, W, a=0.3, b=0.7, c=2.0):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    return (a * xx + b * yy + c)[None, None].astype(np.float32)


def test_roi_align_is_exact_on_linear_images():
    """bilinear sampling reproduces a linear function exactly and the sampling grid is symmetric in the bin, so the output is
    the function at the bin centre"""
    img = _ramp(40, 50)
    rois = np.array([[0, 5.0, 4.0, 25.0, 20.0], [0, 2.5, 3.25, 40.0, 30.5]], dtype=np.float32)
    for sampling in (2, 0, 3):
        out = fo.roi_align(img, rois, 8, 10, sampling)
        for r, (_, x1, y1, x2, y2) in enumerate(rois):
            bw, bh = (x2 - x1) / 10, (y2 - y1) / 8
            ex = 0.3 * (x1 + (np.arange(10) + 0.5) * bw)[None, :] + 0.7 * (y1 + (np.arange(8) + 0.5) * bh)[:, None] + 2.0
            assert np.abs(out[r, 0] - ex).max() < 2e-5


def test_roi_align_borders_and_minimum_size():
    img = np.ones((1, 2, 16, 16), dtype=np.float32)
    out = fo.roi_align(img, np.array([[0, -40.0, -40.0, -20.0, -20.0]], dtype=np.float32), 4, 4, 2)
    assert (out == 0).all()                                       # samples further than one pixel outside contribute zero
    out = fo.roi_align(img, np.array([[0, 3.0, 3.0, 3.0, 3.0]], dtype=np.float32), 4, 4, 2)
    assert np.allclose(out, 1.0)                                  # aligned=False: a degenerate RoI is widened to 1x1
    half = fo.roi_align(img, np.array([[0, -8.0, 0.0, 8.0, 16.0]], dtype=np.float32), 1, 2, 2)
    assert half[0, 0, 0, 0] < 0.3 and np.isclose(half[0, 0, 0, 1], 1.0)   # left half of the RoI hangs out of the image
    two = np.stack([np.zeros((1, 8, 8), np.float32), np.ones((1, 8, 8), np.float32)])
    out = fo.roi_align(two, np.array([[1, 1.0, 1.0, 6.0, 6.0], [0, 1.0, 1.0, 6.0, 6.0]], dtype=np.float32), 3, 3, 2)
    assert np.allclose(out[0], 1.0) and np.allclose(out[1], 0.0)  # first RoI column selects the image


def test_cubic_resize_known_answers():
    from freepose_amd.src.pipeline.refiner_utils import _cubic_coeffs, cubic_resize
    c = _cubic_coeffs(np.array([0.5, 0.0, 0.25], dtype=np.float32))
    assert np.allclose(c[0], [-0.09375, 0.59375, 0.59375, -0.09375]) and np.allclose(c[1], [0, 1, 0, 0])
    assert np.allclose(c.sum(-1), 1.0)
    x = np.random.default_rng(0).random((37, 37)).astype(np.float32)
    assert np.array_equal(cubic_resize(x, (37, 37)), x)           # scale 1: taps (0,1,0,0)
    assert np.allclose(cubic_resize(np.full((518, 518), 0.7, np.float32), (37, 37)), 0.7, atol=1e-6)
    m = np.zeros((518, 518), np.float32)
    m[140:420, 70:350] = 1                                        # block-aligned square: 20 x 20 patches of 14 px
    r = cubic_resize(m, (37, 37)) > 0.5
    assert r.sum() == 400 and r[10:30, 5:25].all()
    # 14x decimation reads the 4x4 neighbourhood of the patch centre: a centre-only blob survives, a corner blob does not
    m = np.zeros((518, 518), np.float32)
    m[14 * 3 + 5:14 * 3 + 9, 14 * 4 + 5:14 * 4 + 9] = 1
    m[14 * 8:14 * 8 + 3, 14 * 9:14 * 9 + 3] = 1
    r = cubic_resize(m, (37, 37)) > 0.5
    assert r[3, 4] and not r[8, 9]


def test_crop_box_and_intrinsics_arithmetic():
    from freepose_amd.src.pipeline import refiner_utils as ru
    K = torch.tensor([[600.0, 0, 320.0], [0, 600.0, 240.0], [0, 0, 1]])
    # whole 518x518 frame as the crop: focal lengths unchanged, principal point shifted by the reference's -0.5 convention
    nk = ru.update_K_with_crop(K, torch.tensor([[0.0, 0.0, 518.0, 518.0]]), 518, 518)[0]
    assert torch.allclose(nk[0, 0], torch.tensor(600.0)) and torch.allclose(nk[0, 2], torch.tensor(319.5))
    # a half-size crop doubles the focal length and maps the crop centre to the render centre
    nk = ru.update_K_with_crop(K, torch.tensor([[190.5, 110.5, 449.5, 369.5]]), 518, 518)[0]
    assert torch.allclose(nk[0, 0], torch.tensor(1200.0)) and torch.allclose(nk[1, 1], torch.tensor(1200.0))
    assert abs(float(nk[0, 2]) - 258.5) < 1.01 and abs(float(nk[1, 2]) - 258.5) < 1.01


def test_confidence_threshold_is_the_top_quantile_bin_edge():
    from freepose_amd.src.pipeline.estimators.tracking_refiner import TrackingRefiner
    tr = TrackingRefiner.__new__(TrackingRefiner)                 # no ViT needed for the host-side statistic
    sims = np.concatenate([np.linspace(0.01, 1.0, 1000), -np.ones(50), np.zeros(50)]).reshape(11, 100)
    thr = tr._get_threshold_for_confidence(sims, top_quantile=0.2)
    assert 0.78 <= thr <= 0.81
    assert abs((sims > thr).sum() / 1000 - 0.2) < 0.03


def test_refiner_host_arithmetic_matches_the_reference(golden_dir):
    """tests/golden/refiner.npz was produced by the reference's own TrackingRefiner._crop_image /
    refiner_utils.update_K_with_crop / _get_threshold_for_confidence (oracle/gen_golden_refiner.py): the sampled object points,
    crop boxes, RoIs handed to roi_align, cropped intrinsics and thresholds must match bit for bit."""
    import types

    from freepose_amd.src.pipeline import refiner_utils as ru
    from freepose_amd.src.pipeline.estimators.tracking_refiner import TrackingRefiner
    g = np.load(golden_dir / "refiner.npz")
    mesh = types.SimpleNamespace(vertices=g["verts"])
    pts = TrackingRefiner._sample_points(mesh)
    expect = np.pad(g["verts"][g["pick"]], ((0, 0), (0, 1)), constant_values=1.0).astype(np.float32)
    assert np.array_equal(pts.numpy(), expect)
    K = torch.from_numpy(g["K"]).view(3, 3).float()
    for T, bbox, new_K, rois in zip(g["transforms"], g["bboxes"], g["new_K"], g["rois"]):
        Ts = torch.from_numpy(T).view(1, 4, 4).float()
        boxes = ru.crop_boxes(Ts, pts, K, 518, 518)
        assert np.array_equal(boxes.numpy()[0], bbox)
        assert np.array_equal(torch.cat([torch.zeros((1, 1)), boxes], 1).numpy(), rois)
        assert np.array_equal(ru.update_K_with_crop(K, boxes, 518, 518).numpy()[0], new_K)
    assert np.array_equal(ru.update_K_with_crop(K, torch.from_numpy(g["direct_boxes"]), 518, 518).numpy(), g["direct_new_K"])
    tr = TrackingRefiner.__new__(TrackingRefiner)
    for q, thr in zip((0.2, 0.05, 0.5), g["thresholds"]):
        assert float(tr._get_threshold_for_confidence(g["sims"], top_quantile=q)) == float(thr)


def test_cubic_resize_agrees_with_an_independent_implementation():
    """the restated cv2 INTER_CUBIC (float32 path) against torch's own bicubic (`F.interpolate(mode="bicubic",
    align_corners=False)`: the same published kernel — a = -0.75, half-pixel centres, replicated border — written independently in
    ATen).  cv2 itself is not installable here; what stays STATED is only cv2's u8 fixed-point path (not used: the validity mask is
    resized as float32, refiner_utils.py:165-167)."""
    import torch.nn.functional as F
    from freepose_amd.src.pipeline.refiner_utils import cubic_resize
    rng = np.random.default_rng(3)
    for (H, W), (dh, dw) in (((518, 518), (37, 37)), ((37, 37), (100, 61)), ((120, 75), (33, 90)), ((64, 64), (64, 64))):
        x = rng.standard_normal((H, W)).astype(np.float32)
        mine = cubic_resize(x, (dw, dh))
        ref = F.interpolate(torch.from_numpy(x)[None, None], size=(dh, dw), mode="bicubic", align_corners=False)[0, 0].numpy()
        assert mine.shape == ref.shape == (dh, dw)
        # (ATen forms the source coordinate in float32, the restatement — like cv2 — in float64: the tap weights differ in their
        #  last places, the results by < 1e-5 of the data range; a wrong kernel constant, border rule or centre convention is > 1e-2)
        assert np.abs(mine - ref).max() < 2e-5 * max(1.0, np.abs(x).max()), ((H, W), (dh, dw), np.abs(mine - ref).max())
    # the thresholded validity mask (what pose_confidence uses) is the same set of patches under both
    m = (rng.random((518, 518)) > 0.55).astype(np.float32)
    m[100:300, 150:420] = 1
    a = cubic_resize(m, (37, 37)) > 0.5
    b = F.interpolate(torch.from_numpy(m)[None, None], size=(37, 37), mode="bicubic", align_corners=False)[0, 0].numpy() > 0.5
    assert np.array_equal(a, b)


def _roi_align_second_opinion(img, rois, PH, PW, s):
    """RoIAlign (aligned=False, sampling_ratio s > 0) written a second time from the published definition, on torch's grid sampler:
    sample point (iy, ix) of bin (ph, pw) at  y1 + (ph + (iy + .5) / s) bin_h,  x1 + (pw + (ix + .5) / s) bin_w;  a point more than one
    pixel outside the image counts 0, any other is clamped into [0, H-1] x [0, W-1] and read bilinearly; a bin is the mean of its s*s
    points.  (grid_sample with align_corners=True and border padding is exactly "clamp, then bilinear between the neighbours".)"""
    import torch.nn.functional as F
    img = torch.from_numpy(img).double()
    N, C, H, W = img.shape
    out = torch.zeros((len(rois), C, PH, PW), dtype=torch.float64)
    for r, (b, x1, y1, x2, y2) in enumerate(np.asarray(rois, dtype=np.float64)):
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        ys = y1 + (torch.arange(PH * s, dtype=torch.float64) + 0.5) * (rh / PH / s)
        xs = x1 + (torch.arange(PW * s, dtype=torch.float64) + 0.5) * (rw / PW / s)
        valid = ((ys >= -1) & (ys <= H))[:, None] & ((xs >= -1) & (xs <= W))[None, :]
        gy = (ys.clamp(0, H - 1) / max(H - 1, 1)) * 2 - 1
        gx = (xs.clamp(0, W - 1) / max(W - 1, 1)) * 2 - 1
        grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], dim=-1)[None]            # [1, PH*s, PW*s, (x, y)]
        samp = F.grid_sample(img[int(b)][None], grid, mode="bilinear", padding_mode="border", align_corners=True)[0]
        samp = samp * valid[None]
        out[r] = samp.reshape(C, PH, s, PW, s).mean(dim=(2, 4))
    return out.numpy()


def test_roi_align_agrees_with_a_second_formulation():
    """the oracle's RoIAlign (restated from torchvision's CPU kernel, which is not installable here) against an
    independently written torch formulation of the published operator, on random images and RoIs that include boxes hanging out of
    the image, sub-pixel boxes and the refiner's own call shape (one 3 x 518 x 518 crop, sampling_ratio 2)."""
    rng = np.random.default_rng(11)
    img = rng.standard_normal((2, 3, 60, 80)).astype(np.float32)
    rois = np.array([[0, 5.2, 4.1, 61.7, 48.3], [1, -6.0, -3.5, 30.0, 25.0], [0, 40.0, 30.0, 95.0, 75.0], [1, 10.3, 10.3, 10.6, 10.4],
                     [0, -30.0, 5.0, -2.0, 40.0], [1, 0.0, 0.0, 80.0, 60.0]], dtype=np.float32)
    for (PH, PW, s) in ((7, 9, 2), (16, 16, 3), (1, 1, 2)):
        got = fo.roi_align(img, rois, PH, PW, s)
        ref = _roi_align_second_opinion(img, rois, PH, PW, s)
        assert np.abs(got - ref).max() < 1e-4, (PH, PW, s, np.abs(got - ref).max())     # float32 (oracle) vs float64 sample coordinates
    big = rng.random((1, 3, 240, 320)).astype(np.float32)
    roi = np.array([[0, 37.25, 12.5, 291.75, 203.0]], dtype=np.float32)
    got = fo.roi_align(big, roi, 518, 518, 2)
    assert np.abs(got - _roi_align_second_opinion(big, roi, 518, 518, 2)).max() < 1e-4
