"""SURVEY §8(f)-3 on the MI355X: fp_roi_align and the ambient-5 raster bit-exact against the oracle, and
TrackingRefiner.pose_confidence on a planted pose."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sampling", [2, 0, 3])
def test_roi_align_bit_exact(sampling):
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    rng = np.random.default_rng(7)
    img = rng.random((2, 3, 97, 131)).astype(np.float32)
    rois = np.array([[0, 5.5, 4.25, 60.0, 71.5], [1, -20.0, -10.0, 50.0, 40.0], [1, 100.0, 60.0, 180.0, 140.0],
                     [0, 10.0, 10.0, 10.0, 10.0], [0, -300.0, -300.0, -200.0, -200.0], [1, 0.0, 0.0, 131.0, 97.0]], dtype=np.float32)
    got = ops.roi_align(torch.from_numpy(img), torch.from_numpy(rois), (23, 31), sampling_ratio=sampling).cpu().numpy()
    ref = fo.roi_align(img, rois, 23, 31, sampling)
    assert np.array_equal(got, ref)


def test_roi_align_518_crop_of_a_frame():
    """the shape TrackingRefiner uses: one 3 x 720 x 1280 frame -> 3 x 518 x 518"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    img = np.random.default_rng(1).random((1, 3, 720, 1280)).astype(np.float32)
    rois = np.array([[0, 400.3, 150.7, 900.9, 651.3]], dtype=np.float32)
    got = ops.roi_align(torch.from_numpy(img), torch.from_numpy(rois), (518, 518)).cpu().numpy()
    assert np.array_equal(got, fo.roi_align(img, rois, 518, 518, 2))


def test_raster_ambient_factor_bit_exact():
    import bench
    from freepose_amd import ops
    from freepose_amd.src.pipeline.retrieval.renderer import grid_poses
    from oracle import fp_oracle as fo
    v, f, c = bench.synthetic_mesh(3)
    poses = np.array(grid_poses(3)).astype(np.float32)
    mesh = ops.Mesh(v, f, c).set_ambient(5.0)
    rgb, depth = ops.rasterize(mesh, torch.from_numpy(poses), 0.25, 800.0, 790.0, 250.0, 262.0, 518, 518)
    r_rgb, r_depth = fo.rasterize(v, f, c, poses, 0.25, 800.0, 790.0, 250.0, 262.0, 518, 518, ambient=5.0)
    assert np.array_equal(rgb.cpu().numpy(), r_rgb) and np.array_equal(depth.cpu().numpy(), r_depth)
    rgb2, _ = ops.rasterize(ops.Mesh(v, f, c), torch.from_numpy(poses), 0.25, 800.0, 790.0, 250.0, 262.0, 518, 518)
    assert (rgb.cpu().numpy().astype(int) >= rgb2.cpu().numpy().astype(int)).all() and not torch.equal(rgb, rgb2)


def test_pose_confidence_prefers_the_true_pose():
    """photo = the object rendered (ambient 5) under a known pose into a 640x480 frame.  pose_confidence must be high where
    the render is valid for the true pose and clearly lower for a wrong rotation; n_inliers_per_pose must rank them."""
    import bench
    from freepose_amd import ops
    from freepose_amd.mesh_io import TriMesh
    from freepose_amd.src.pipeline.retrieval.renderer import grid_poses
    from src.pipeline.estimators.tracking_refiner import TrackingRefiner      # the reference's import path
    v, f, c = bench.synthetic_mesh(4)
    mesh = TriMesh(v * 0.25, f, c)
    K = np.array([[600.0, 0, 320.0], [0, 600.0, 240.0], [0, 0, 1]])
    poses = np.array(grid_poses(40))
    T_true, T_wrong = poses[3].copy(), poses[3].copy()
    T_wrong[:3, :3] = poses[29][:3, :3]
    for T in (T_true, T_wrong):
        T[:3, 3] = (0.05, -0.03, 1.2)
    dm = ops.Mesh(mesh.vertices, mesh.faces, mesh.vertex_colors).set_ambient(5.0)
    photo, _ = ops.rasterize(dm, torch.from_numpy(T_true[None].astype(np.float32)), 1.0, 600.0, 600.0, 320.0, 240.0, 640, 480)
    photo = photo[0].cpu().numpy()
    tr = TrackingRefiner(dino_model="dinov2_vitb14_reg", seed=4)
    assert (tr.image_size, tr.patch_size, tr.feats_size) == (518, 14, 37)
    good = tr.pose_confidence(mesh, photo, K, T_true)
    bad = tr.pose_confidence(mesh, photo, K, T_wrong)
    assert good.shape == (37, 37) and good.dtype == np.float32
    assert (good != 0).sum() > 200                                  # the object fills the 1.4x crop
    assert good[good != 0].mean() > 0.9 and good[good != 0].mean() > bad[bad != 0].mean() + 0.05
    n, thr = tr.n_inliers_per_pose(mesh, [photo, photo], K, [T_true, T_wrong])
    assert n.shape == (2,) and n[0] > n[1] and 0 < thr < 1
    # the pairs of a clip share ViT calls (windows of 16 pairs = 32 crops): pair for pair the confidences of the one-by-one calls
    many = tr.pose_confidences(mesh, [photo] * 5, K, [T_true, T_wrong, T_true, T_wrong, T_true], window=3)
    assert many.shape == (5, 37, 37)
    for i, ref in enumerate((good, bad, good, bad, good)):
        assert np.array_equal(many[i], ref), i
    with pytest.raises(NotImplementedError):
        tr.refine
