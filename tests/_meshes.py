"""Small deterministic meshes / textures for the rasteriser tests (test data only)."""
import numpy as np


def checker_gradient_texture(n=256, cells=8, seed=3):
    """u8 [n,n,3]: a checker of random colours modulated by horizontal / vertical ramps (every texel distinct in value or place)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = rng.integers(0, 256, size=(cells, cells, 3), dtype=np.uint8)
    tex = np.kron(base, np.ones((n // cells, n // cells, 1), np.uint8)).astype(np.int32)
    yy, xx = np.mgrid[0:n, 0:n]
    tex[..., 0] = (tex[..., 0] + xx // 2) % 256
    tex[..., 1] = (tex[..., 1] + yy // 3) % 256
    tex[..., 2] = (tex[..., 2] + (xx + yy) // 5) % 256
    return tex.astype(np.uint8)


def textured_cube():
    """12-triangle cube, half-extent 1, one quad of the texture per face (cross-free atlas: 3 x 2 cells), per-corner uv"""
    corners = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], np.float32)
    quads = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (4, 5, 1, 0), (3, 2, 6, 7)]
    faces, uvs = [], []
    for qi, (a, b, c, d) in enumerate(quads):
        u0, v0 = (qi % 3) / 3.0, (qi // 3) / 2.0
        u1, v1 = u0 + 1 / 3.0, v0 + 0.5
        quv = [(u0, v0), (u1, v0), (u1, v1), (u0, v1)]
        faces += [[a, b, c], [a, c, d]]
        uvs += [[quv[0], quv[1], quv[2]], [quv[0], quv[2], quv[3]]]
    return corners, np.array(faces, np.int32), np.array(uvs, np.float32)


def write_textured_obj(folder, name, tex, kd=(1.0, 1.0, 1.0)):
    """<folder>/<name>.obj + .mtl + texture PNG of the textured cube (OBJ `vt` per corner, one material)"""
    from pathlib import Path
    from PIL import Image
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    v, f, uv = textured_cube()
    Image.fromarray(tex, "RGB").save(folder / f"{name}_tex.png")
    (folder / f"{name}.mtl").write_text(f"newmtl m0\nKd {kd[0]} {kd[1]} {kd[2]}\nmap_Kd {name}_tex.png\n")
    lines = [f"mtllib {name}.mtl"] + [f"v {x} {y} {z}" for x, y, z in v]
    flat = uv.reshape(-1, 2)
    lines += [f"vt {a} {b}" for a, b in flat]
    lines.append("usemtl m0")
    for i, tri in enumerate(f):
        lines.append("f " + " ".join(f"{tri[k] + 1}/{3 * i + k + 1}" for k in range(3)))
    (folder / f"{name}.obj").write_text("\n".join(lines) + "\n")
    return folder / f"{name}.obj"
