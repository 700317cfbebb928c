"""GPU parity (through the C ABI) for the round-2 rasteriser work: per-fragment UV texture sampling (both visibility
strategies, both shading rules, bit for bit against oracle/fp_oracle.c), the vertex-stage export against the oracle and the
reference-projection golden, the render_templates round trip on a textured OBJ, and the WebTemplateDataset mirror against the
golden produced by the REFERENCE's loader on the same synthetic shard."""
import hashlib
import tarfile

import numpy as np
import pytest
import torch

from tests._meshes import checker_gradient_texture, textured_cube, write_textured_obj

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _poses(n, seed=0):
    from oracle import fp_oracle as fo
    P = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    P[:, :3, :3] = fo.generate_rotations(n)
    P[:, :3, 3] = [0, 0, 1.1]
    rng = np.random.Generator(np.random.PCG64(seed))
    P[n - 1, :3, 3] = [0.25, -0.15, 0.8]         # partly off-screen, strongly minified / magnified mix
    P[0, :3, 3] = [0.0, 0.0, 0.45]               # very close: magnification (bilinear weights matter)
    _ = rng
    return P


@pytest.mark.parametrize("W,H,shade,ambient,kd", [(420, 420, 1, 2.0, None), (420, 420, 0, 2.0, None), (518, 518, 1, 5.0, (0.9, 0.5, 1.0)),
                                                  (300, 200, 0, 1.0, (1.0, 1.0, 0.25)), (704, 480, 1, 2.0, None)])
def test_textured_cube_bit_exact_both_strategies(W, H, shade, ambient, kd):
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    v, f, uv = textured_cube()
    tex = checker_gradient_texture(256)
    poses = _poses(6)
    fx = 600.0 * W / 420
    rgb_o, d_o = fo.rasterize(v, f, None, poses, 0.25, fx, fx, W / 2, H / 2, W, H, ambient=ambient, shade=shade, uv=uv, texture=tex, kd=kd)
    mesh = ops.Mesh(v, f, uv=uv, texture=tex, kd=kd).set_ambient(ambient).set_shading(shade)
    try:
        for mode in (1, 0):
            ops.set_option("raster_tiled", mode)
            rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(poses), 0.25, fx, fx, W / 2, H / 2, W, H)
            assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32)), f"depth differs (tiled={mode})"
            diff = rgb_g.cpu().numpy() != rgb_o
            assert not diff.any(), f"rgb differs at {int(diff.sum())} values (tiled={mode})"
    finally:
        ops.set_option("raster_tiled", -1)
    cov = d_o > 0
    assert cov.mean() > 0.1
    # the texture is really on screen: many distinct colours, not a flat fill
    assert len(np.unique(rgb_o[cov].reshape(-1, 3), axis=0)) > 500


@pytest.mark.parametrize("filt", [1, 0])
def test_minified_texture_mip_chain_bit_exact(filt):
    """1024^2 noise-on-checker texture on the cube seen from 0.45 m to 6 m: levels 0 .. 5 of the mip chain are in play (and the
    far views are a handful of pixels); both visibility strategies against the oracle, trilinear and level-0-only filtering"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    v, f, uv = textured_cube()
    rng = np.random.Generator(np.random.PCG64(21))
    tex = (checker_gradient_texture(1024, cells=64, seed=5).astype(np.int32) // 2 + rng.integers(0, 128, size=(1024, 1024, 3))).astype(np.uint8)
    poses = _poses(6)
    for i, z in enumerate([0.45, 1.1, 2.0, 3.5, 6.0, 0.8]):
        poses[i, :3, 3] = [0.02 * i, -0.01 * i, z]
    rgb_o, d_o = fo.rasterize(v, f, None, poses, 0.25, 600, 600, 210, 210, 420, 420, uv=uv, texture=tex, filter=filt)
    mesh = ops.Mesh(v, f, uv=uv, texture=tex).set_filter(filt)
    try:
        for mode in (1, 0):
            ops.set_option("raster_tiled", mode)
            rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(poses), 0.25, 600, 600, 210, 210, 420, 420)
            assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32))
            diff = rgb_g.cpu().numpy() != rgb_o
            assert not diff.any(), f"rgb differs at {int(diff.sum())} values (tiled={mode}, filter={filt})"
    finally:
        ops.set_option("raster_tiled", -1)
    if filt == 1:   # minification really smooths: the far view's colours vary far less than the unfiltered lookup's
        rgb_n, _ = fo.rasterize(v, f, None, poses[3:4], 0.25, 600, 600, 210, 210, 420, 420, uv=uv, texture=tex, filter=0)
        c = d_o[3] > 0
        assert c.sum() > 500
        gx1 = np.abs(np.diff(rgb_o[3].astype(np.int32), axis=1))[c[:, 1:] & c[:, :-1]].mean()
        gx0 = np.abs(np.diff(rgb_n[0].astype(np.int32), axis=1))[c[:, 1:] & c[:, :-1]].mean()
        assert gx1 < 0.6 * gx0, (gx1, gx0)


def test_textured_dense_mesh_and_repeat_wrap_bit_exact():
    """displaced icosphere (20 480 triangles) with spherical uv scaled x3 (REPEAT wrap in play, seam triangles span a period)"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    from tests.test_gpu_retrieval import _icosphere
    v, f = _icosphere(5)
    v = (v * (1 + 0.15 * np.sin(5 * v[:, :1]))).astype(np.float32)
    f = f.astype(np.int32)
    p = v / np.linalg.norm(v, axis=1, keepdims=True)
    vuv = np.stack([np.arctan2(p[:, 1], p[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(p[:, 2], -1, 1)) / np.pi], axis=1)
    uv = (3.0 * vuv[f.reshape(-1)]).reshape(-1, 3, 2).astype(np.float32)
    tex = checker_gradient_texture(128, cells=16, seed=8)
    poses = _poses(4)
    poses[0, :3, 3] = [0, 0, 1.1]
    rgb_o, d_o = fo.rasterize(v, f, None, poses, 0.25, 600, 600, 210, 210, 420, 420, uv=uv, texture=tex)
    mesh = ops.Mesh(v, f, uv=uv, texture=tex)
    try:
        for mode in (1, 0):
            ops.set_option("raster_tiled", mode)
            rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(poses), 0.25, 600, 600, 210, 210, 420, 420)
            assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32))
            assert np.array_equal(rgb_g.cpu().numpy(), rgb_o)
    finally:
        ops.set_option("raster_tiled", -1)


def test_dense_mesh_any_face_order_on_the_tiled_path():
    """81 920 triangles (the bench's mesh size; rounds 1-5 sent anything above 32 768 to the global atomics buffer) with the faces in
    RANDOM order: the tiled strategy (Morton-ordered 64-triangle chunks, LDS hit lists) is the default one up to 131 072 triangles and
    gives the oracle's bits; so do the extents / boxes of the fused epilogue; and the face order does not matter (ids ride in the key)"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    import bench
    v, f, col = bench.synthetic_mesh(6)
    assert len(f) == 81920
    rng = np.random.default_rng(5)
    f = np.ascontiguousarray(f[rng.permutation(len(f))]).astype(np.int32)
    poses = _poses(3)
    poses[1, :3, 3] = [0.05, -0.02, 0.6]
    rgb_o, d_o = fo.rasterize(v, f, col, poses, 0.25, 600, 600, 210, 210, 420, 420)
    ext_o = fo.depth_extents(d_o, 600, 600, 210, 210)
    mesh = ops.Mesh(v, f, col)
    rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(poses), 0.25, 600, 600, 210, 210, 420, 420)          # default strategy
    assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32)) and np.array_equal(rgb_g.cpu().numpy(), rgb_o)
    try:
        for mode in (1, 0):
            ops.set_option("raster_tiled", mode)
            rgb_f, d_f, ext_f, box_f = ops.rasterize_extents(mesh, torch.from_numpy(poses), 0.25, 600, 600, 210, 210, 420, 420, want_depth=bool(mode))
            assert np.array_equal(rgb_f.cpu().numpy(), rgb_o) and np.array_equal(ext_f.cpu().numpy(), ext_o)
            assert np.array_equal(box_f.cpu().numpy(), ext_o[:, :4].astype(np.int32))
    finally:
        ops.set_option("raster_tiled", -1)


def test_million_triangle_mesh_both_strategies():
    """1 310 720 triangles (icosphere level 8: most triangles cover no pixel centre at all; 20 480 chunks = ten rounds of the tile kernel's
    mask scan, 16-bit hit ids relative to the round): the tiled strategy forced on, the global one and the oracle agree bit for bit, and so
    do the fused extents"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    import bench
    v, f, col = bench.synthetic_mesh(8)
    assert len(f) == 1310720
    f = f.astype(np.int32)
    poses = _poses(2)
    rgb_o, d_o = fo.rasterize(v, f, col, poses, 0.25, 600, 600, 210, 210, 420, 420)
    ext_o = fo.depth_extents(d_o, 600, 600, 210, 210)
    mesh = ops.Mesh(v, f, col)
    try:
        for mode in (1, 0):
            ops.set_option("raster_tiled", mode)
            rgb_g, d_g, ext_g, box_g = ops.rasterize_extents(mesh, torch.from_numpy(poses), 0.25, 600, 600, 210, 210, 420, 420, want_depth=True)
            assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32)) and np.array_equal(rgb_g.cpu().numpy(), rgb_o), mode
            assert np.array_equal(ext_g.cpu().numpy(), ext_o) and np.array_equal(box_g.cpu().numpy(), ext_o[:, :4].astype(np.int32))
    finally:
        ops.set_option("raster_tiled", -1)


def test_texel_pattern_known_answer_on_device():
    """same construction as the oracle's CPU known-answer test: texel centres land on pixel centres -> exact texels"""
    from freepose_amd import ops
    rng = np.random.Generator(np.random.PCG64(9))
    tex = rng.integers(0, 256, size=(8, 8, 3), dtype=np.uint8)
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    corner = {0: (0.0, 1.0), 1: (1.0, 1.0), 2: (1.0, 0.0), 3: (0.0, 0.0)}
    uv = np.array([[corner[i] for i in tri] for tri in f], np.float32)
    P = np.eye(4, dtype=np.float32)
    P[2, 3] = 1.0
    mesh = ops.Mesh(v, f, uv=uv, texture=tex).set_ambient(1.0).set_shading(0)
    rgb, d = ops.rasterize(mesh, torch.from_numpy(P[None]), 0.1, 200.0, 200.0, 20.0, 20.0, 40, 40)
    assert bool((d > 0).all())
    assert np.array_equal(rgb[0].cpu().numpy()[2::5, 2::5], tex)


def test_project_vertices_matches_oracle_and_reference_golden(golden_dir):
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    g = np.load(golden_dir / "projection.npz", allow_pickle=True)
    for name in g["names"]:
        K, scale = g[f"{name}_K"], float(g[f"{name}_scale"])
        verts, poses = g[f"{name}_verts"], g[f"{name}_poses"].astype(np.float32)
        faces = np.array([[0, 1, 2]], np.int32)
        mesh = ops.Mesh(verts, faces)
        xy_g, zc_g = ops.project_vertices(mesh, torch.from_numpy(poses), scale, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        xy_o, zc_o = fo.project_vertices(verts, poses, scale, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        assert np.array_equal(xy_g.cpu().numpy(), xy_o)
        assert np.array_equal(zc_g.cpu().numpy().view(np.uint32), zc_o.view(np.uint32))
        ref_xy, ref_z = g[f"{name}_xy"], g[f"{name}_zeye"]
        vis = (ref_z > 0.06) & (np.abs(ref_xy).max(axis=-1) < 20000)
        err = np.abs(xy_g.cpu().numpy().astype(np.float64) / 256.0 - ref_xy)[vis]
        assert err.max() <= 0.5 / 256 + 1e-3 * np.abs(ref_xy[vis]).max() / 1000


def test_render_templates_roundtrip_textured_obj(tmp_path, monkeypatch):
    """OBJ + MTL + PNG texture -> scripts.render_templates -> shard -> WebTemplateDataset: views equal a direct textured
    render, which equals the oracle's textured render of the same arrays"""
    from freepose_amd.mesh_io import load_obj
    from freepose_amd.src.dataloader.template import WebTemplateDataset
    from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer
    from oracle import fp_oracle as fo
    from scripts import render_templates
    from tests.test_gpu_pipeline import ops_crop_identity
    n_views = 8
    tex = checker_gradient_texture(128)
    write_textured_obj(tmp_path / "mesh_cache" / "cube_t", "cube_t", tex, kd=(1.0, 0.8, 0.9))
    (tmp_path / "list.txt").write_text("cube_t\n")
    monkeypatch.delenv("SLURM_ARRAY_TASK_ID", raising=False)
    tar_path = render_templates.run(["--filelist", str(tmp_path / "list.txt"), "--mesh_root", str(tmp_path / "mesh_cache"),
                                     "--datasets_root", str(tmp_path / "datasets"), "--shards_folder", "sh", "--n_views", str(n_views)])
    with tarfile.open(tar_path) as tar:
        assert "cubet_0.rgb.png" in tar.getnames()
    (tmp_path / "list.csv").write_text("model_name\ncube_t\n")
    s = WebTemplateDataset(str(tmp_path / "datasets" / "sh"), str(tmp_path / "list.csv"), crop=False, n_views=n_views)[0]
    mesh = load_obj(tmp_path / "mesh_cache" / "cube_t" / "cube_t.obj")
    assert mesh.uv is not None and mesh.texture is not None
    r = MeshRenderer(n_views)
    direct = r.render(mesh, scale=0.25)
    assert torch.equal(s["templates"].cpu(), ops_crop_identity(direct.rgb).cpu())
    rgb_o, d_o = fo.rasterize(mesh.vertices, mesh.faces, None, np.array(r.mesh_poses, np.float32), 0.25, 600, 600, 210, 210, 420, 420,
                              uv=mesh.uv, texture=mesh.texture, kd=mesh.kd)
    assert np.array_equal(direct.rgb.cpu().numpy(), rgb_o)
    assert np.array_equal(direct.depth.cpu().numpy().view(np.uint32), d_o.view(np.uint32))
    assert len(np.unique(rgb_o[d_o > 0].reshape(-1, 3), axis=0)) > 300


def test_web_template_dataset_matches_reference_golden(tmp_path, golden_dir):
    """the reference's WebTemplateDataset.get_template_by_name (src/dataloader/template.py:46-99) was run on the shard that
    tests/synth_shard.py writes (oracle/gen_golden_r2.py); the mirror must return the same tensors: crops bit for bit
    (incl. bbox_extend, the < 100 px fallback square and empty views), masks, depths, boxes, dtypes, names"""
    from freepose_amd.src.dataloader.template import WebTemplateDataset
    from tests import synth_shard
    g = np.load(golden_dir / "template_dataset.npz", allow_pickle=True)
    names = synth_shard.write_shard(tmp_path)
    assert names == [str(n) for n in g["names"]]
    for ext in (0, 0.05):
        ds = WebTemplateDataset(str(tmp_path / "shards"), str(tmp_path / "mesh_cache.csv"), bbox_extend=ext)
        e = ds.get_template_by_name(names[1])
        t = e["templates"].cpu().numpy()
        key = f"e{ext}"
        assert t.shape == (600, 3, 420, 420) and str(e["templates"].dtype) == str(g["templates_dtype"])
        assert np.array_equal(t[:3], g[f"{key}_templates_first3"])
        assert np.array_equal(t[::40, :, ::7, ::7], g[f"{key}_templates_sample"])
        bad = [i for i in range(600) if sha(t[i]) != str(g[f"{key}_templates_sha"][i])]
        assert not bad, f"crops differ from the reference at views {bad[:10]} (bbox_extend={ext})"
        if ext == 0:
            m = e["masks"].cpu().numpy()
            d = e["depths"].cpu().numpy()
            assert m.dtype == np.bool_ and str(e["depths"].dtype) == str(g["depths_dtype"])
            assert [sha(x) for x in m] == [str(x) for x in g["masks_sha"]]
            assert [sha(x) for x in d] == [str(x) for x in g["depths_sha"]]
            assert np.array_equal(m.reshape(600, -1).sum(1), g["mask_counts"])
            assert np.array_equal(e["bboxes"].cpu().numpy(), g["bboxes"])
            assert np.array_equal(e["intrinsic"].numpy(), g["intrinsic"]) and e["intrinsic"].dtype == torch.int64
            assert e["model_name"] == str(g["model_name"]) and e["tar_file"] == str(g["tar_file"])
    nc = WebTemplateDataset(str(tmp_path / "shards"), str(tmp_path / "mesh_cache.csv"), crop=False).get_template_by_name(names[0])
    assert [sha(x) for x in nc["templates"].cpu().numpy()[:20]] == [str(x) for x in g["nocrop_templates_sha"]]
    # ---- the device-resident store: a hit returns a FRESH dict over the SAME device tensors (nothing is decoded), still
    # equal to the reference golden; the LRU holds `cache_meshes` entries; cache_meshes = 0 is the reference's decode-per-call
    ds = WebTemplateDataset(str(tmp_path / "shards"), str(tmp_path / "mesh_cache.csv"), bbox_extend=0, cache_meshes=1)
    e1 = ds.get_template_by_name(names[1])
    spent = ds.decode_seconds
    e2 = ds.get_template_by_name(names[1])
    assert ds.decode_seconds == spent and e2 is not e1
    assert all(e2[k].data_ptr() == e1[k].data_ptr() for k in ("templates", "masks", "depths", "bboxes"))
    e2["model_name"] = "scribbled"                                   # a caller editing its dict does not reach the store
    assert ds.get_template_by_name(names[1])["model_name"] == str(g["model_name"])
    assert not [i for i in range(0, 600, 7) if sha(e2["templates"][i].cpu().numpy()) != str(g["e0_templates_sha"][i])]
    ds.get_template_by_name(names[0])                                # evicts names[1] (one slot)
    assert ds.decode_seconds > spent
    spent = ds.decode_seconds
    e3 = ds.get_template_by_name(names[1])
    assert ds.decode_seconds > spent and torch.equal(e3["templates"], e1["templates"]) and torch.equal(e3["depths"], e1["depths"])
    ds0 = WebTemplateDataset(str(tmp_path / "shards"), str(tmp_path / "mesh_cache.csv"), bbox_extend=0, cache_meshes=0)
    a, b = ds0.get_template_by_name(names[1]), ds0.get_template_by_name(names[1])
    assert a["templates"].data_ptr() != b["templates"].data_ptr() and torch.equal(a["templates"], b["templates"])
    # ---- prefetch (round 5): the host stage of a mesh (tar reads, PNG decode, host->device copy on a side stream) started in the
    # background gives the entry the synchronous load gives, bit for bit, whatever the order of requests; a prefetched mesh is
    # consumed exactly once; prefetching a resident or unknown mesh is a no-op
    dp = WebTemplateDataset(str(tmp_path / "shards"), str(tmp_path / "mesh_cache.csv"), bbox_extend=0, cache_meshes=0)
    dp.prefetch(1)
    dp.prefetch(0)
    dp.prefetch(1)                                                   # already on its way
    dp.prefetch_by_name("no-such-mesh")
    assert set(dp._pending) == {0, 1}
    p1, p0 = dp[1], dp[0]
    assert not dp._pending
    for got, want in ((p1, a), (p0, ds0.get_template_by_name(names[0]))):
        for k in ("templates", "masks", "depths", "bboxes"):
            assert torch.equal(got[k], want[k]), k
        assert got["model_name"] == want["model_name"] and got["tar_file"] == want["tar_file"]
    dr = WebTemplateDataset(str(tmp_path / "shards"), str(tmp_path / "mesh_cache.csv"), bbox_extend=0, cache_meshes=2)
    dr[1]
    dr.prefetch(1)
    assert not dr._pending
    # prefetched but never fetched meshes do not pile up (each done one would pin ~530 MB of device memory): oldest dropped beyond the bound
    db = WebTemplateDataset(str(tmp_path / "shards"), str(tmp_path / "mesh_cache.csv"), bbox_extend=0, cache_meshes=0)
    db.MAX_PENDING = 1
    db.prefetch(0)
    db.prefetch(1)
    assert set(db._pending) == {1}
    for got, want in ((db[0], p0), (db[1], p1)):
        assert torch.equal(got["templates"], want["templates"]) and torch.equal(got["depths"], want["depths"])


@pytest.mark.parametrize("textured", [False, True])
def test_near_plane_straddlers_bit_exact_both_strategies(textured):
    """triangles that straddle z = 0.05 (renderer.py:62-67): the camera sits INSIDE a textured / vertex-coloured cube and next to a floor
    quad that runs from behind it to 3 m ahead — homogeneous-coordinate rasterisation, both visibility strategies, bit for bit against
    the oracle (whose clipping is checked against the analytic answer in tests/test_golden_r2_cpu.py)"""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    v, f, uv = textured_cube()
    floor_v = np.array([[-8, 0.04, -4], [8, 0.04, -4], [8, 0.04, 12], [-8, 0.04, 12]], np.float32)
    floor_f = np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(v)
    floor_uv = np.array([[[0, 0], [1, 0], [1, 1]], [[0, 0], [1, 1], [0, 1]]], np.float32)
    v = np.concatenate([v, floor_v])
    f = np.concatenate([f, floor_f])
    uv = np.concatenate([uv, floor_uv])
    tex = checker_gradient_texture(256)
    col = np.random.default_rng(5).integers(0, 256, size=(len(v), 3), dtype=np.uint8)
    P = np.tile(np.eye(4, dtype=np.float32), (4, 1, 1))
    P[:, :3, :3] = fo.generate_rotations(4)
    P[0, :3, 3] = [0.0, 0.0, 0.10]       # camera inside the (scaled) cube: every face straddles or lies behind
    P[1, :3, 3] = [0.05, -0.02, 0.30]    # a corner pokes through the near plane
    P[2, :3, 3] = [0.0, 0.0, 1.10]       # the ordinary case in the same batch
    P[3, :3, 3] = [0.3, 0.1, 0.02]
    W = H = 420
    kw = dict(uv=uv, texture=tex) if textured else {}
    rgb_o, d_o = fo.rasterize(v, f, None if textured else col, P, 0.25, 600, 600, 210, 210, W, H, **kw)
    mesh = ops.Mesh(v, f, uv=uv, texture=tex) if textured else ops.Mesh(v, f, col)
    try:
        for mode in (1, 0):
            ops.set_option("raster_tiled", mode)
            rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(P), 0.25, 600, 600, 210, 210, W, H)
            assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32)), f"depth differs (tiled={mode})"
            assert np.array_equal(rgb_g.cpu().numpy(), rgb_o), f"rgb differs (tiled={mode})"
    finally:
        ops.set_option("raster_tiled", -1)
    assert (d_o[0] > 0).mean() > 0.5 and d_o[0][d_o[0] > 0].min() > 0.05      # something is drawn in the inside view, nothing nearer than znear
    assert (d_o[2] > 0).mean() > 0.05


@pytest.mark.parametrize("textured", [False, True])
def test_back_face_culling_bit_exact_both_strategies(textured):
    """`cull_faces=True` (renderer.py:63-66, :90-93): same scene as the straddler test (so that both the ordinary and the homogeneous
    path decide facing), both visibility strategies, bit for bit against the oracle; then the MeshRenderer wrapper: a closed
    outward-wound box renders the same either way and the flag does not stick to the cached device mesh."""
    from freepose_amd import ops
    from oracle import fp_oracle as fo
    v, f, uv = textured_cube()
    floor_v = np.array([[-8, 0.04, -4], [8, 0.04, -4], [8, 0.04, 12], [-8, 0.04, 12]], np.float32)
    floor_f = np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(v)
    floor_uv = np.array([[[0, 0], [1, 0], [1, 1]], [[0, 0], [1, 1], [0, 1]]], np.float32)
    v = np.concatenate([v, floor_v])
    f = np.concatenate([f, floor_f])
    uv = np.concatenate([uv, floor_uv])
    tex = checker_gradient_texture(256)
    col = np.random.default_rng(5).integers(0, 256, size=(len(v), 3), dtype=np.uint8)
    P = np.tile(np.eye(4, dtype=np.float32), (6, 1, 1))
    P[:, :3, :3] = fo.generate_rotations(6)
    P[0, :3, 3] = [0.0, 0.0, 0.10]
    P[1, :3, 3] = [0.05, -0.02, 0.30]
    P[2, :3, 3] = [0.0, 0.0, 1.10]
    P[3, :3, 3] = [0.3, 0.1, 0.02]
    P[4, :3, 3] = [0.1, 0.0, 0.9]
    P[5, :3, 3] = [-0.2, 0.1, 1.5]
    W = H = 420
    kw = dict(uv=uv, texture=tex) if textured else {}
    differs = 0
    for faces in (f, f[:, ::-1].copy()):
        uvf = uv[:, ::-1].copy() if faces is not f else uv
        kwf = dict(uv=uvf, texture=tex) if textured else {}
        rgb_o, d_o = fo.rasterize(v, faces, None if textured else col, P, 0.25, 600, 600, 210, 210, W, H, cull=1, **kwf)
        rgb_n, d_n = fo.rasterize(v, faces, None if textured else col, P, 0.25, 600, 600, 210, 210, W, H, **kwf)
        differs += int((d_o != d_n).sum())
        mesh = (ops.Mesh(v, faces, uv=uvf, texture=tex) if textured else ops.Mesh(v, faces, col)).set_cull(1)
        try:
            for mode in (1, 0):
                ops.set_option("raster_tiled", mode)
                rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(P), 0.25, 600, 600, 210, 210, W, H)
                assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32)), f"depth differs (tiled={mode})"
                assert np.array_equal(rgb_g.cpu().numpy(), rgb_o), f"rgb differs (tiled={mode})"
            mesh.set_cull(0)
            rgb_g, d_g = ops.rasterize(mesh, torch.from_numpy(P), 0.25, 600, 600, 210, 210, W, H)
            assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_n.view(np.uint32))
            assert np.array_equal(rgb_g.cpu().numpy(), rgb_n)
        finally:
            ops.set_option("raster_tiled", -1)
        assert (d_o > 0).sum() > 10000
    assert differs > 10000                       # culling removed something in these views (inside views, the floor from below)


def test_mesh_renderer_cull_faces():
    """the reference's MeshRenderer.render / render_from_poses keyword (renderer.py:56, :83)"""
    from freepose_amd import ops
    from freepose_amd.src.pipeline.retrieval.renderer import MeshRenderer
    v, f, _ = textured_cube()                                        # wound inwards (tests/test_golden_r2_cpu.py)
    v = v * np.array([1.0, 0.7, 0.45], np.float32)
    col = np.random.default_rng(2).integers(0, 255, size=(len(v), 3), dtype=np.uint8)
    outward, inward = ops.Mesh(v, f[:, ::-1].copy(), col), ops.Mesh(v, f, col)
    r = MeshRenderer(n_poses=8, resolution=224)
    a = r.render(outward, scale=0.25)
    b = r.render(outward, cull_faces=True, scale=0.25)
    assert torch.equal(a.depth, b.depth) and torch.equal(a.rgb, b.rgb) and int((a.depth > 0).sum()) > 1000
    d0 = r.render_from_poses(inward, r.mesh_poses, scale=0.25)
    c = r.render_from_poses(inward, r.mesh_poses, cull_faces=True, scale=0.25)      # only the far inside walls face the camera
    d = r.render_from_poses(inward, r.mesh_poses, scale=0.25)                       # the flag does not stick to the device mesh
    assert torch.equal(d.depth, d0.depth) and torch.equal(d.rgb, d0.rgb)
    assert torch.allclose(d.depth, a.depth, rtol=1e-5, atol=0)                      # same surface, other vertex order: rounding only
    assert torch.equal(c.depth > 0, d.depth > 0) and bool((c.depth > d.depth).any()) and not bool((c.depth < d.depth).any())
