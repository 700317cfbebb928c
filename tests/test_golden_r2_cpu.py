"""Round-2 pins (no GPU): the oracle's vertex stage against the reference tree's own K -> OpenGL projection, the CSV row format
against the reference's own formatting statements, known answers of the oracle's texture sampling, and the OBJ/MTL reader.
Fixtures: oracle/gen_golden_r2.py -> tests/golden/{projection,csv_rows}.npz."""
import io

import numpy as np
import pytest

from oracle import fp_oracle as fo
from tests._meshes import checker_gradient_texture, textured_cube, write_textured_obj


def _g(golden_dir, name):
    return np.load(golden_dir / name, allow_pickle=True)


# ---- a11: vertex stage vs bop_toolkit_lib/renderer_py.py:186-231 + renderer.py:37-41 ------------------------
def test_vertex_stage_matches_reference_projection(golden_dir):
    g = _g(golden_dir, "projection.npz")
    for name in g["names"]:
        K, (W, H), scale = g[f"{name}_K"], g[f"{name}_WH"], float(g[f"{name}_scale"])
        xy, zc = fo.project_vertices(g[f"{name}_verts"], g[f"{name}_poses"], scale, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        ref_xy, ref_z = g[f"{name}_xy"], g[f"{name}_zeye"]
        vis = (ref_z > 0.06) & (np.abs(ref_xy).max(axis=-1) < 20000)       # in front of the near plane (0.05), not clamped
        assert vis.mean() > 0.8
        # window coordinates: pixel centres at +0.5, x right, y down; 24.8 fixed point => |err| <= half a sub-pixel + fp32 noise
        err = np.abs(xy.astype(np.float64) / 256.0 - ref_xy)[vis]
        assert err.max() <= 0.5 / 256 + 1e-3 * np.abs(ref_xy[vis]).max() / 1000, (name, err.max())
        # eye depth is the camera-frame z (linear, metres), not a normalised depth-buffer value
        assert np.allclose(zc[vis], ref_z[vis], rtol=2e-6, atol=1e-7)
        assert (xy[ref_z < 0.04] == 0).all()                                 # at / behind the near plane: dropped


def test_rendered_depth_is_eye_depth_at_projected_vertex(golden_dir):
    """a fronto-parallel quad at z = 1.3 rendered by the oracle: depth image == 1.3 inside, 0 outside, and its silhouette
    ends where the reference projection puts the quad's corners (pixel p is covered iff its centre p + .5 is inside)"""
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    P = np.eye(4, dtype=np.float32)
    P[:3, 3] = [0.013, -0.021, 1.3]
    fx, fy, cx, cy, W, H = 600.0, 590.0, 210.0, 200.0, 420, 400
    _, d = fo.rasterize(v, f, None, P[None], 0.25, fx, fy, cx, cy, W, H)
    u0, u1 = fx * (-0.25 + 0.013) / 1.3 + cx, fx * (0.25 + 0.013) / 1.3 + cx
    v0, v1 = fy * (-0.25 - 0.021) / 1.3 + cy, fy * (0.25 - 0.021) / 1.3 + cy
    cols = np.where((d[0] > 0).any(axis=0))[0]
    rows = np.where((d[0] > 0).any(axis=1))[0]
    assert cols.min() == int(np.ceil(u0 - 0.5)) and cols.max() == int(np.floor(u1 - 0.5 - 1e-9))
    assert rows.min() == int(np.ceil(v0 - 0.5)) and rows.max() == int(np.floor(v1 - 0.5 - 1e-9))
    assert np.allclose(d[0][d[0] > 0], 1.3, rtol=1e-6)


# ---- a12: CSV rows vs scripts/dino_inference.py:113-130 and scripts/dino_inference_video.py:160-182 -----------
def test_csv_rows_match_reference_format(golden_dir):
    import pandas as pd
    from freepose_amd.scripts.dino_inference import CSV_COLUMNS, pose_row
    g = _g(golden_dir, "csv_rows.npz")
    meshes, scales = [str(m) for m in g["meshes"]], [float(s) for s in g["scales"]]
    rows = [pose_row("48", 7 + i, meshes[i], g["scores"][i][0], g["tco"][i], g["bbox"][i], scales[i]) for i in range(4)]
    buf = io.StringIO()
    pd.DataFrame(rows, columns=CSV_COLUMNS).to_csv(buf, index=False, header=True)
    assert buf.getvalue() == str(g["image_csv"])
    rows = [pose_row(0, fr, meshes[o], g["scores"][2 * fr + o][0], g["tco"][2 * fr + o], g["bbox"][2 * fr + o], scales[o],
                     t_scale=1, time_value=-1) for fr in range(2) for o in range(2)]
    buf = io.StringIO()
    pd.DataFrame(rows, columns=CSV_COLUMNS).to_csv(buf, index=False, header=True)
    assert buf.getvalue() == str(g["video_csv"])
    # the bop_toolkit reader expects exactly 9 comma-separated fields (bop_toolkit_lib/inout.py:297-347)
    for line in buf.getvalue().strip().splitlines():
        assert len(line.split(",")) == 9


# ---- a11: texture sampling known answers (oracle; the HIP kernels are compared bit for bit in test_gpu_raster_textured.py) ----
def _quad(z=1.0):
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    P = np.eye(4, dtype=np.float32)
    P[2, 3] = z
    return v, f, P


def test_texel_pattern_survives_linear_shading():
    """8 x 8 texture on a fronto-parallel quad covering exactly 8 x 8 blocks of 5 x 5 pixels: block centres coincide with
    texel centres, so bilinear filtering returns the texel itself; OBJ convention: v = 1 is the image's first row, and the
    camera looks along +z with y down, so object +y is image-down."""
    rng = np.random.Generator(np.random.PCG64(9))
    tex = rng.integers(0, 256, size=(8, 8, 3), dtype=np.uint8)
    v, f, P = _quad(1.0)
    # quad spans x,y in [-0.5, 0.5] * ... : scale 0.1 -> [-0.1, 0.1] m at z = 1 with f = 200 -> 40 px wide, centred at 20
    uv_corner = {0: (0.0, 1.0), 1: (1.0, 1.0), 2: (1.0, 0.0), 3: (0.0, 0.0)}      # top-left vertex (x-, y-) gets v = 1
    uv = np.array([[uv_corner[i] for i in tri] for tri in f], np.float32)
    rgb, d = fo.rasterize(v, f, None, P[None], 0.1, 200.0, 200.0, 20.0, 20.0, 40, 40, ambient=1.0, shade=0, uv=uv, texture=tex)
    assert (d[0] > 0).all()
    centres = rgb[0][2::5, 2::5]
    assert np.array_equal(centres, tex)
    # wrap = REPEAT: shifting every coordinate by whole periods changes nothing
    rgb2, _ = fo.rasterize(v, f, None, P[None], 0.1, 200.0, 200.0, 20.0, 20.0, 40, 40, ambient=1.0, shade=0, uv=uv + [2.0, -3.0], texture=tex)
    assert np.array_equal(rgb2, rgb)
    # reversing a triangle's winding (orientation swap inside the rasteriser) must carry the corner attributes along
    f2, uv2 = f.copy(), uv.copy()
    f2[1] = f2[1][[0, 2, 1]]
    uv2[1] = uv2[1][[0, 2, 1]]
    rgb3, _ = fo.rasterize(v, f2, None, P[None], 0.1, 200.0, 200.0, 20.0, 20.0, 40, 40, ambient=1.0, shade=0, uv=uv2, texture=tex)
    assert np.array_equal(rgb3, rgb)
    # between two texel centres the filter is the linear blend
    a, b = tex[3, 2].astype(np.float64), tex[3, 3].astype(np.float64)
    mid = rgb[0][3 * 5 + 2, 2 * 5 + 2 + 2].astype(np.float64)          # 2 px right of texel (3,2)'s centre: weight 0.4
    assert np.abs(mid - (0.6 * a + 0.4 * b)).max() <= 1.0


def test_mip_levels_and_level_of_detail_known_answers():
    """Trilinear minification of the textured path (oracle; the device kernels are bit-compared with it in the gpu tests).
    1-texel checker of 0 / 255: every level >= 1 is the constant 128 ((0+255+0+255+2)>>2).  A fronto-parallel quad that maps n
    texels onto n/2 pixels has rho = 2, level of detail exactly 1 -> every pixel is 128; mapped onto 4n pixels it is magnified ->
    level 0, identical to the unfiltered lookup; at rho = sqrt(2) the result is the half-way blend of level 0 and 128."""
    n = 64
    yy, xx = np.mgrid[0:n, 0:n]
    tex = np.where(((xx + yy) % 2)[..., None] == 0, np.uint8(255), np.uint8(0)).repeat(3, axis=2).astype(np.uint8)
    v, f, P = _quad(1.0)
    uv_corner = {0: (0.0, 1.0), 1: (1.0, 1.0), 2: (1.0, 0.0), 3: (0.0, 0.0)}
    uv = np.array([[uv_corner[i] for i in tri] for tri in f], np.float32)

    def render(px, filt):      # the quad spans px x px pixels of a (px + 8)^2 image
        W = px + 8
        return fo.rasterize(v, f, None, P[None], 0.1, px / 0.2, px / 0.2, W / 2, W / 2, W, W, ambient=1.0, shade=0, uv=uv, texture=tex,
                            filter=filt)

    rgb, d = render(n // 2, 1)
    cov = d[0] > 0
    assert cov.sum() == (n // 2) ** 2 and (rgb[0][cov] == 128).all()
    big1, dm = render(4 * n, 1)
    big0, _ = render(4 * n, 0)
    assert np.array_equal(big1, big0) and (dm[0] > 0).sum() == (4 * n) ** 2
    # rho = sqrt(2): 64 texels on 45.25 px is not integral; use 2 : sqrt(2) via anisotropy-free scaling of the image instead:
    # 64 texels onto 45 px -> rho = 1.4222, log2 = 0.5082 -> weight of level 1 = 0.5082 (+- the 2e-5 polynomial)
    rgbh, dh = render(45, 1)
    lvl0, _ = render(45, 0)
    covh = dh[0] > 0
    w = np.log2(64 / 45.0)
    want = (1 - w) * lvl0[0][covh].astype(np.float64) + w * 128.0
    assert np.abs(rgbh[0][covh].astype(np.float64) - want).max() <= 1.0
    # white noise at rho = 4: level 2 averages 16 texels per sample; the level-0 bilinear lookup (pixel centres on texel corners) only 4
    rng = np.random.Generator(np.random.PCG64(4))
    tex = rng.integers(0, 256, size=(n, n, 3), dtype=np.uint8)
    a1, da = render(n // 4, 1)
    a0, _ = render(n // 4, 0)
    c = da[0] > 0
    assert a1[0][c].astype(np.float64).std() < 0.6 * a0[0][c].astype(np.float64).std()      # 16 vs 4 texels per sample


def test_gamma_shading_rule_and_material_factor():
    dec, thr = fo.shade_tables()
    assert dec[0] == 0 and abs(dec[255] - 1) < 1e-6 and abs(dec[128] - 0.21586) < 1e-4 and (np.diff(dec) > 0).all()
    assert (np.diff(thr[1:]) > 0).all()
    tex = np.zeros((4, 4, 3), np.uint8)
    tex[..., 0], tex[..., 1], tex[..., 2] = 64, 128, 250
    v, f, P = _quad(1.0)
    uv = np.array([[(0.1, 0.1), (0.9, 0.1), (0.9, 0.9)], [(0.1, 0.1), (0.9, 0.9), (0.1, 0.9)]], np.float32)
    for amb, kd in ((2.0, None), (5.0, None), (2.0, (0.5, 1.0, 0.25))):
        rgb, _ = fo.rasterize(v, f, None, P[None], 0.1, 200.0, 200.0, 20.0, 20.0, 40, 40, ambient=amb, shade=1, uv=uv, texture=tex, kd=kd)
        k = np.ones(3) if kd is None else np.array(kd)
        lin = np.array([dec[64], dec[128], dec[250]], np.float64) * k * amb
        want = np.clip(np.round(255.0 * np.clip(lin, 0, 1) ** (1 / 2.2)), 0, 255)
        assert np.abs(rgb[0, 20, 20].astype(np.float64) - want).max() <= 1, (amb, kd)
    # vertex colours go through the same transfer: 2 * (64/255) -> (0.50)^(1/2.2) * 255 = 186
    col = np.full((4, 3), 64, np.uint8)
    rgb, _ = fo.rasterize(v, f, col, P[None], 0.1, 200.0, 200.0, 20.0, 20.0, 40, 40, ambient=2.0, shade=1)
    assert abs(int(rgb[0, 20, 20, 0]) - round(255 * (2 * 64 / 255) ** (1 / 2.2))) <= 1
    rgb, _ = fo.rasterize(v, f, col, P[None], 0.1, 200.0, 200.0, 20.0, 20.0, 40, 40, ambient=2.0, shade=0)
    assert int(rgb[0, 20, 20, 0]) == 128


def test_texture_lookup_is_perspective_correct():
    """a quad tilted about the vertical axis: the screen-space midpoint between its left and right edges shows the texel at
    the perspective-correct u (< 0.5 towards the far edge), not the affine one"""
    n = 64
    tex = np.zeros((n, n, 3), np.uint8)
    tex[..., 0] = np.arange(n)[None, :] * 4              # red encodes the column
    v, f, _ = _quad()
    uv = np.array([[(0, 1), (1, 1), (1, 0)], [(0, 1), (1, 0), (0, 0)]], np.float32)
    th = np.deg2rad(55.0)
    P = np.eye(4, dtype=np.float32)
    P[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    P[2, 3] = 1.0
    fx = 300.0
    rgb, d = fo.rasterize(v, f, None, P[None], 0.25, fx, fx, 100.0, 100.0, 200, 200, ambient=1.0, shade=0, uv=uv, texture=tex)
    row = 100
    cols = np.where(d[0, row] > 0)[0]
    px = (cols.min() + cols.max()) // 2
    # analytic: pixel centre -> ray -> intersection with the quad's plane -> object x in [-0.25, 0.25] -> u
    xn = (px + 0.5 - 100.0) / fx
    # points on the quad: X = R @ (s, y, 0) + t ; x/z = xn  =>  s*cos = xn*(1 - s*sin)
    s = xn / (np.cos(th) + xn * np.sin(th))
    u = (s / 0.25 + 1) / 2
    want = (u * n - 0.5) * 4
    assert abs(float(rgb[0, row, px, 0]) - want) <= 4.5
    affine_u = 0.5
    assert abs(u - affine_u) > 0.05                      # the test distinguishes the two


# ---- mesh reader -------------------------------------------------------------------------------------------------
def test_load_obj_keeps_per_corner_uv_texture_and_kd(tmp_path):
    from freepose_amd.mesh_io import load_obj, mesh_appearance, mesh_signature
    tex = checker_gradient_texture(64)
    path = write_textured_obj(tmp_path / "m", "cube", tex, kd=(0.8, 0.6, 1.0))
    m = load_obj(path)
    v, f, uv = textured_cube()
    assert np.array_equal(m.faces, f) and np.allclose(m.vertices, v)
    assert m.uv.shape == (12, 3, 2) and np.allclose(m.uv, uv, atol=1e-6)       # seams keep their own coordinates per corner
    assert np.array_equal(m.texture, tex) and np.allclose(m.kd, [0.8, 0.6, 1.0])
    app = mesh_appearance(m)
    assert set(app) == {"uv", "texture", "kd"}
    sig = mesh_signature(m)
    m.apply_scale(0.25)
    assert mesh_signature(m) != sig                      # in-place scaling invalidates device-mesh caches


def test_load_obj_multi_material_atlas(tmp_path):
    from PIL import Image
    from freepose_amd.mesh_io import load_obj
    d = tmp_path / "mm"
    d.mkdir()
    t0 = np.full((8, 8, 3), (200, 10, 10), np.uint8)
    Image.fromarray(t0, "RGB").save(d / "a.png")
    (d / "m.mtl").write_text("newmtl red\nKd 1 1 1\nmap_Kd a.png\nnewmtl blue\nKd 0.0 0.0 1.0\n")
    (d / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                             "usemtl red\nf 1/1 2/2 3/3\nusemtl blue\nf 1 3 4\n")
    m = load_obj(d / "m.obj")
    assert m.texture is not None and m.uv.shape == (2, 3, 2)
    h, w = m.texture.shape[:2]

    def texel(u, v):
        return m.texture[min(h - 1, int((1 - v) * h)), min(w - 1, int(u * w))]
    assert tuple(texel(*m.uv[0].mean(axis=0))) == (200, 10, 10)
    assert tuple(texel(*m.uv[1].mean(axis=0))) == (0, 0, 255)


def test_trimesh_like_texture_visual_is_honoured():
    """objects shaped like trimesh.Trimesh with TextureVisuals (per-vertex uv + material.image) must not render white"""
    import types
    from PIL import Image
    from freepose_amd.mesh_io import mesh_appearance
    v, f, uv = textured_cube()
    vv = v[f.reshape(-1)]                                 # trimesh duplicates vertices per (v, vt) pair
    ff = np.arange(len(vv), dtype=np.int32).reshape(-1, 3)
    img = Image.fromarray(checker_gradient_texture(32), "RGB")
    mat = types.SimpleNamespace(image=img, diffuse=np.array([255, 128, 255, 255], np.uint8))
    mesh = types.SimpleNamespace(vertices=vv, faces=ff, visual=types.SimpleNamespace(uv=uv.reshape(-1, 2), material=mat))
    app = mesh_appearance(mesh)
    assert np.allclose(app["uv"], uv) and app["texture"].shape == (32, 32, 3)
    assert np.allclose(app["kd"], [1.0, 128 / 255, 1.0])


def test_near_plane_clipping_known_answers():
    """renderer.py:62-67 renders with znear = 0.05 (pyrender clips there).  A floor quad that runs from BEHIND the camera to 3 m in
    front of it (both triangles straddle the near plane): depth must equal the analytic ray / plane intersection wherever that lies in
    (0.05, 3], nothing may be drawn where it is nearer than the plane, and vertex colours must interpolate perspective-correctly."""
    from oracle import fp_oracle as fo
    W = H = 420
    fx = fy = 600.0
    cx = cy = 210.0
    y0 = 0.01
    v = np.array([[-2, y0, -1], [2, y0, -1], [2, y0, 3], [-2, y0, 3]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    col = np.array([[0, 10, 0], [0, 250, 0], [255, 250, 0], [255, 10, 0]], np.uint8)      # red = 255 (z + 1) / 4, green = 10 + 60 (x + 2)
    pose = np.eye(4, dtype=np.float32)[None]
    rgb, depth = fo.rasterize(v, f, col, pose, 1.0, fx, fy, cx, cy, W, H, ambient=1.0, shade=0)
    py = np.arange(H)[:, None] + 0.5
    px = np.arange(W)[None, :] + 0.5
    with np.errstate(divide="ignore", invalid="ignore"):
        z = np.where(py > cy, y0 * fy / (py - cy), np.inf) * np.ones((1, W))
    x = (px - cx) / fx * z
    inside = (z > 0.05) & (z <= 3.0) & (np.abs(x) <= 2.0)
    # coverage: every clearly-inside pixel is drawn, every clearly-outside pixel is empty (one-pixel band at the far edge left open:
    # the far edge is an ordinary projected edge, snapped to 1/256 px)
    band = np.abs(z - 3.0) < 0.5
    got = depth[0] > 0
    assert np.array_equal(got[~band], inside[~band])
    assert got[:, :].sum() > 40000
    rows = np.nonzero(got.any(axis=1))[0]
    assert rows.max() == 329, rows.max()                   # z(329) = 0.0502 > 0.05 >= z(330) = 0.0498: clipped exactly at the near plane
    sel = got & inside
    assert np.allclose(depth[0][sel], z[sel], rtol=2e-5)
    red = 255.0 * (z + 1.0) / 4.0
    green = 10.0 + 60.0 * (x + 2.0)
    assert np.abs(rgb[0][..., 0][sel].astype(np.float64) - np.minimum(255, np.floor(red[sel] + 0.5))).max() <= 1
    assert np.abs(rgb[0][..., 1][sel].astype(np.float64) - np.minimum(255, np.floor(green[sel] + 0.5))).max() <= 1
    # a triangle entirely behind the plane draws nothing; one entirely in front is untouched by the new path
    v2 = v.copy()
    v2[:, 2] = [-3, -3, -0.5, -0.5]
    assert not (fo.rasterize(v2, f, col, pose, 1.0, fx, fy, cx, cy, W, H)[1] > 0).any()


def test_back_face_culling_known_answers():
    """renderer.py:63-66 / :90-93: `cull_faces=True` drops pyrender's SKIP_CULL_FACES flag, so OpenGL's default culling applies
    (front = counter-clockwise seen from outside).  Known answers: a closed, outward-wound box seen from outside renders the same
    with and without culling; a single triangle is drawn from its front side only; the near-plane path (homogeneous rasterisation)
    uses the same sign as the ordinary path."""
    from oracle import fp_oracle as fo
    from tests._meshes import textured_cube
    W = H = 128
    fx = fy = 200.0
    cx = cy = 64.0
    v, f, _ = textured_cube()
    f = f[:, ::-1].copy()                                     # the test cube is wound inwards; make it outward (CCW from outside)
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    assert ((n * (v[f].mean(1) - v.mean(0))).sum(1) > 0).all()
    col = (np.arange(v.shape[0] * 3).reshape(-1, 3) * 9 % 256).astype(np.uint8)
    poses = []
    rng = np.random.Generator(np.random.PCG64(4))
    from scipy.spatial.transform import Rotation as Rot
    for _ in range(6):
        P = np.eye(4, dtype=np.float32)
        P[:3, :3] = Rot.random(random_state=int(rng.integers(1 << 30))).as_matrix()
        P[:3, 3] = [0.0, 0.0, 4.0]
        poses.append(P)
    poses = np.array(poses)
    a = fo.rasterize(v, f, col, poses, 1.0, fx, fy, cx, cy, W, H)
    b = fo.rasterize(v, f, col, poses, 1.0, fx, fy, cx, cy, W, H, cull=1)
    assert (a[1] > 0).sum() > 1000
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # the same box with every face reversed: only the far inside walls would be front-facing -> a different (deeper) image
    c = fo.rasterize(v, f[:, ::-1].copy(), col, poses, 1.0, fx, fy, cx, cy, W, H, cull=1)
    assert np.array_equal(c[1] > 0, a[1] > 0) and (c[1][a[1] > 0] >= a[1][a[1] > 0]).all() and (c[1] > a[1]).any()
    # one triangle, camera looks down +z with y down: (0,0) (1,0) (0,1) at z = 2 runs clockwise on screen as seen by the camera, i.e.
    # counter-clockwise from BEHIND: it is a back face; reversed it is a front face
    tri = np.array([[0, 0, 2], [0.5, 0, 2], [0, 0.5, 2]], np.float32)
    one = np.array([[0, 1, 2]], np.int32)
    eye = np.eye(4, dtype=np.float32)[None]
    tc = np.full((3, 3), 200, np.uint8)
    assert (fo.rasterize(tri, one, tc, eye, 1.0, fx, fy, cx, cy, W, H)[1] > 0).sum() > 500
    assert not (fo.rasterize(tri, one, tc, eye, 1.0, fx, fy, cx, cy, W, H, cull=1)[1] > 0).any()
    assert (fo.rasterize(tri, one[:, ::-1].copy(), tc, eye, 1.0, fx, fy, cx, cy, W, H, cull=1)[1] > 0).sum() > 500
    # floor strip under the camera, split into a part that straddles the near plane and a part that does not: one winding, one verdict
    y0 = 0.05
    fl = np.array([[-1, y0, -1], [1, y0, -1], [1, y0, 1], [-1, y0, 1], [1, y0, 3], [-1, y0, 3]], np.float32)
    quads = np.array([[0, 1, 2], [0, 2, 3], [3, 2, 4], [3, 4, 5]], np.int32)
    fc = np.full((6, 3), 180, np.uint8)
    full = fo.rasterize(fl, quads, fc, eye, 1.0, fx, fy, cx, cy, W, H)[1] > 0
    near_part = fo.rasterize(fl, quads[:2], fc, eye, 1.0, fx, fy, cx, cy, W, H)[1] > 0
    far_part = fo.rasterize(fl, quads[2:], fc, eye, 1.0, fx, fy, cx, cy, W, H)[1] > 0
    assert near_part.sum() > 500 and far_part.sum() > 100
    seen = [(fo.rasterize(fl, q, fc, eye, 1.0, fx, fy, cx, cy, W, H, cull=1)[1] > 0) for q in (quads, quads[:, ::-1].copy())]
    assert sorted([int(s.sum()) for s in seen]) == [0, int(full.sum())]
    assert np.array_equal(seen[0] | seen[1], full)
