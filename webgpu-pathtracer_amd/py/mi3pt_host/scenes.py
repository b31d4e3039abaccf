"""Synthetic scenes for tests and the benchmark.

`demo_scene()` restates the reference's default scene (src/main.ts:36-75): a 5x5 plane
rotated -90 deg about X, a 0.8^3 red box at (0, 0.4, 0.5) and a radius-0.5 32x32 sphere at
(0, 0.5, -0.5), flattened to world-space triangles the way RaytracePass.updateScene does
(src/passes/raytrace.ts:406-502).  The geometry generators follow three.js r171's
PlaneGeometry / BoxGeometry / SphereGeometry (vertex order, normals, index order), which
is third-party code outside the reference tree (yarn.lock:1380) -- restated from the
published algorithm, parity unpinned.  All arithmetic is float64 like JavaScript;
attribute arrays are rounded to float32 where three.js stores Float32Array.

The larger scenes (`dragon_class_scene`, `forest_scene`) and `synthetic_env` are this
project's own seeded generators (no counterpart in the reference).
"""
import math

import numpy as np

from . import layout

# ---------------------------------------------------------------------------------
# three.js-style geometry (positions / normals Float32, index list)
# ---------------------------------------------------------------------------------


def plane_geometry(width=1.0, height=1.0, width_segments=1, height_segments=1):
    width_half, height_half = width / 2, height / 2
    grid_x, grid_y = int(math.floor(width_segments)), int(math.floor(height_segments))
    grid_x1, grid_y1 = grid_x + 1, grid_y + 1
    segment_width, segment_height = width / grid_x, height / grid_y
    vertices, normals, indices = [], [], []
    for iy in range(grid_y1):
        y = iy * segment_height - height_half
        for ix in range(grid_x1):
            x = ix * segment_width - width_half
            vertices.append((x, -y, 0.0))
            normals.append((0.0, 0.0, 1.0))
    for iy in range(grid_y):
        for ix in range(grid_x):
            a = ix + grid_x1 * iy
            b = ix + grid_x1 * (iy + 1)
            c = (ix + 1) + grid_x1 * (iy + 1)
            d = (ix + 1) + grid_x1 * iy
            indices += [a, b, d, b, c, d]
    return (np.array(vertices, np.float32), np.array(normals, np.float32), np.array(indices, np.int64))


def box_geometry(width=1.0, height=1.0, depth=1.0, ws=1, hs=1, ds=1):
    vertices, normals, indices = [], [], []
    state = {"n": 0}

    def build_plane(u, v, w, udir, vdir, pw, ph, pd, grid_x, grid_y):
        seg_w, seg_h = pw / grid_x, ph / grid_y
        w_half, h_half, d_half = pw / 2, ph / 2, pd / 2
        grid_x1, grid_y1 = grid_x + 1, grid_y + 1
        counter = 0
        for iy in range(grid_y1):
            y = iy * seg_h - h_half
            for ix in range(grid_x1):
                x = ix * seg_w - w_half
                vec = [0.0, 0.0, 0.0]
                vec[u], vec[v], vec[w] = x * udir, y * vdir, d_half
                vertices.append(tuple(vec))
                nrm = [0.0, 0.0, 0.0]
                nrm[w] = 1.0 if pd > 0 else -1.0
                normals.append(tuple(nrm))
                counter += 1
        base = state["n"]
        for iy in range(grid_y):
            for ix in range(grid_x):
                a = base + ix + grid_x1 * iy
                b = base + ix + grid_x1 * (iy + 1)
                c = base + (ix + 1) + grid_x1 * (iy + 1)
                d = base + (ix + 1) + grid_x1 * iy
                indices.extend([a, b, d, b, c, d])
        state["n"] += counter

    X, Y, Z = 0, 1, 2
    build_plane(Z, Y, X, -1, -1, depth, height, width, ds, hs)    # px
    build_plane(Z, Y, X, 1, -1, depth, height, -width, ds, hs)    # nx
    build_plane(X, Z, Y, 1, 1, width, depth, height, ws, ds)      # py
    build_plane(X, Z, Y, 1, -1, width, depth, -height, ws, ds)    # ny
    build_plane(X, Y, Z, 1, -1, width, height, depth, ws, hs)     # pz
    build_plane(X, Y, Z, -1, -1, width, height, -depth, ws, hs)   # nz
    return (np.array(vertices, np.float32), np.array(normals, np.float32), np.array(indices, np.int64))


def sphere_geometry(radius=1.0, width_segments=32, height_segments=16):
    width_segments = max(3, int(math.floor(width_segments)))
    height_segments = max(2, int(math.floor(height_segments)))
    phi_start, phi_length, theta_start, theta_length = 0.0, math.pi * 2, 0.0, math.pi
    theta_end = min(theta_start + theta_length, math.pi)
    index = 0
    grid, vertices, normals, indices = [], [], [], []
    for iy in range(height_segments + 1):
        row = []
        v = iy / height_segments
        for ix in range(width_segments + 1):
            u = ix / width_segments
            x = -radius * math.cos(phi_start + u * phi_length) * math.sin(theta_start + v * theta_length)
            y = radius * math.cos(theta_start + v * theta_length)
            z = radius * math.sin(phi_start + u * phi_length) * math.sin(theta_start + v * theta_length)
            vertices.append((x, y, z))
            length = math.sqrt(x * x + y * y + z * z) or 1.0     # Vector3.normalize = * (1 / length)
            inv = 1.0 / length
            normals.append((x * inv, y * inv, z * inv))
            row.append(index)
            index += 1
        grid.append(row)
    for iy in range(height_segments):
        for ix in range(width_segments):
            a, b = grid[iy][ix + 1], grid[iy][ix]
            c, d = grid[iy + 1][ix], grid[iy + 1][ix + 1]
            if iy != 0 or theta_start > 0:
                indices += [a, b, d]
            if iy != height_segments - 1 or theta_end < math.pi:
                indices += [b, c, d]
    return (np.array(vertices, np.float32), np.array(normals, np.float32), np.array(indices, np.int64))


# ---------------------------------------------------------------------------------
# three.js-style transforms (column-major 4x4 in a flat list `te`, like Matrix4.elements)
# ---------------------------------------------------------------------------------

def quaternion_from_axis_angle(axis, angle):
    half = angle / 2
    s = math.sin(half)
    return (axis[0] * s, axis[1] * s, axis[2] * s, math.cos(half))


def compose_matrix(position=(0.0, 0.0, 0.0), quaternion=(0.0, 0.0, 0.0, 1.0), scale=(1.0, 1.0, 1.0)):
    """Matrix4.compose"""
    x, y, z, w = quaternion
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz = x * x2, x * y2, x * z2
    yy, yz, zz = y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    sx, sy, sz = scale
    return [
        (1 - (yy + zz)) * sx, (xy + wz) * sx, (xz - wy) * sx, 0.0,
        (xy - wz) * sy, (1 - (xx + zz)) * sy, (yz + wx) * sy, 0.0,
        (xz + wy) * sz, (yz - wx) * sz, (1 - (xx + yy)) * sz, 0.0,
        position[0], position[1], position[2], 1.0,
    ]


def normal_matrix(te):
    """Matrix3.getNormalMatrix(m4) = setFromMatrix4(m4).invert().transpose(); returns
    column-major 3x3 elements."""
    n11, n21, n31 = te[0], te[1], te[2]
    n12, n22, n32 = te[4], te[5], te[6]
    n13, n23, n33 = te[8], te[9], te[10]
    t11 = n33 * n22 - n32 * n23
    t12 = n32 * n13 - n33 * n12
    t13 = n23 * n12 - n22 * n13
    det = n11 * t11 + n21 * t12 + n31 * t13
    if det == 0:
        return [0.0] * 9
    det_inv = 1 / det
    inv = [
        t11 * det_inv, (n31 * n23 - n33 * n21) * det_inv, (n32 * n21 - n31 * n22) * det_inv,
        t12 * det_inv, (n33 * n11 - n31 * n13) * det_inv, (n31 * n12 - n32 * n11) * det_inv,
        t13 * det_inv, (n21 * n13 - n23 * n11) * det_inv, (n22 * n11 - n21 * n12) * det_inv,
    ]
    # transpose
    return [inv[0], inv[3], inv[6], inv[1], inv[4], inv[7], inv[2], inv[5], inv[8]]


def flatten_mesh(geometry, matrix_world, material_index):
    """raytrace.ts:425-502 for one indexed mesh.  Returns (positions, normals,
    material indices): (n,3,3) float64, (n,3,3) float64, (n,) int."""
    verts, norms, index = geometry
    e = matrix_world
    m = normal_matrix(e)
    v = verts.astype(np.float64)
    n = norms.astype(np.float64)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    # Vector3.applyMatrix4
    w = 1 / (e[3] * x + e[7] * y + e[11] * z + e[15])
    px = (e[0] * x + e[4] * y + e[8] * z + e[12]) * w
    py = (e[1] * x + e[5] * y + e[9] * z + e[13]) * w
    pz = (e[2] * x + e[6] * y + e[10] * z + e[14]) * w
    # Vector3.applyMatrix3(normalMatrix).normalize()
    nx0, ny0, nz0 = n[:, 0], n[:, 1], n[:, 2]
    nx = m[0] * nx0 + m[3] * ny0 + m[6] * nz0
    ny = m[1] * nx0 + m[4] * ny0 + m[7] * nz0
    nz = m[2] * nx0 + m[5] * ny0 + m[8] * nz0
    length = np.sqrt(nx * nx + ny * ny + nz * nz)
    length[length == 0] = 1.0
    inv = 1 / length
    nx, ny, nz = nx * inv, ny * inv, nz * inv
    world_p = np.stack([px, py, pz], axis=1)
    world_n = np.stack([nx, ny, nz], axis=1)
    tri = index.reshape(-1, 3)
    return world_p[tri], world_n[tri], np.full(len(tri), material_index, np.int64)


class Scene:
    """What RaytracePass.updateScene leaves on the device + the camera and knobs."""

    def __init__(self, positions, normals, material_index, materials, name=""):
        self.name = name
        self.positions = positions            # (n,3,3) float64 world space
        self.normals = normals
        self.material_index = material_index
        self.materials = materials            # list of dicts
        self.triangles = layout.pack_triangles(positions, normals, material_index)
        self.material_bytes = layout.pack_materials(materials)
        self.nodes = None                     # filled by build_bvh()
        # camera (main.ts:38-39): fov 45 at (0,1,4) looking at the origin (OrbitControls target)
        self.camera = dict(position=(0.0, 1.0, 4.0), target=(0.0, 0.0, 0.0), fov=45.0,
                           focalDistance=1.0, aperture=0.0)

    def camera_direction(self):
        p, t = np.array(self.camera["position"], np.float64), np.array(self.camera["target"], np.float64)
        d = t - p
        return d / math.sqrt(float(d @ d))

    def build_bvh(self, nthreads=0):
        from . import capi
        self.nodes = capi.host_build_bvh_f64(self.positions, nthreads)
        return self.nodes


WHITE = dict(color=(1.0, 1.0, 1.0), roughness=1.0, metalness=0.02, specularColor=(1.0, 1.0, 1.0))
RED = dict(color=(1.0, 0.05, 0.05), roughness=1.0, metalness=0.0, specularColor=(1.0, 1.0, 1.0))


def _ground_plane():
    q = quaternion_from_axis_angle((1.0, 0.0, 0.0), -math.pi / 2)      # plane.rotateX(-PI / 2)
    return flatten_mesh(plane_geometry(5, 5), compose_matrix(quaternion=q), 0)


def demo_scene():
    """main.ts:36-75: 2 + 12 + 1984 = 1998 triangles, materials [white, red]."""
    parts = [_ground_plane()]
    parts.append(flatten_mesh(box_geometry(0.8, 0.8, 0.8), compose_matrix(position=(0.0, 0.4, 0.5)), 1))
    parts.append(flatten_mesh(sphere_geometry(0.5, 32, 32), compose_matrix(position=(0.0, 0.5, -0.5)), 0))
    pos = np.concatenate([p[0] for p in parts])
    nrm = np.concatenate([p[1] for p in parts])
    mat = np.concatenate([p[2] for p in parts])
    return Scene(pos, nrm, mat, [WHITE, RED], "demo")


# ---------------------------------------------------------------------------------
# own generators
# ---------------------------------------------------------------------------------

def synthetic_env(width=layout.ENV_WIDTH, height=layout.ENV_HEIGHT, sun_dir=(0.45, 0.55, 0.70),
                  sun_radiance=60.0):
    """Analytic sky gradient + ground + sun disk, rgba32float, row 0 = top of the sky
    (the -Y +X file order the reference's RGBELoader produces).  Fixed formula, fp64
    evaluated then rounded to fp32."""
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    theta = (0.5 - v) * math.pi                  # elevation: +pi/2 at row 0
    phi = (u - 0.5) * 2 * math.pi                # matches uv.x = phi / 2pi + 0.5, phi = atan2(x, z)
    ct, st = np.cos(theta)[:, None], np.sin(theta)[:, None]
    dx, dy, dz = np.sin(phi)[None, :] * ct, st + 0 * phi[None, :], np.cos(phi)[None, :] * ct
    t = np.clip(dy, 0.0, 1.0)
    horizon = np.array([0.85, 0.90, 1.00])
    zenith = np.array([0.20, 0.38, 0.85])
    ground = np.array([0.22, 0.20, 0.18])
    sky = horizon[None, None, :] * (1 - t[..., None] ** 0.5) + zenith[None, None, :] * (t[..., None] ** 0.5)
    img = np.where((dy > 0)[..., None], sky, ground[None, None, :] * (1.0 + 0.5 * np.clip(dy, -1, 0)[..., None]))
    s = np.array(sun_dir, np.float64)
    s = s / math.sqrt(float(s @ s))
    cosang = dx * s[0] + dy * s[1] + dz * s[2]
    sun = np.clip((cosang - 0.9990) / (0.9997 - 0.9990), 0.0, 1.0)
    glow = np.clip(cosang, 0.0, 1.0) ** 64
    img = img + sun[..., None] * sun_radiance * np.array([1.0, 0.93, 0.82]) + glow[..., None] * 0.6
    out = np.ones((height, width, 4), np.float32)
    out[..., :3] = img.astype(np.float32)
    return out


def _smooth_normals(verts, faces):
    fn = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
    vn = np.zeros_like(verts)
    for k in range(3):
        np.add.at(vn, faces[:, k], fn)
    ln = np.sqrt((vn * vn).sum(1))
    ln[ln == 0] = 1.0
    return vn / ln[:, None]


def displaced_blob(segments, seed=7, octaves=5):
    """A closed, bumpy, star-shaped surface: a lat/long sphere grid whose radius is
    displaced by seeded sinusoidal noise.  2*segments^2 - 2*segments triangles."""
    rng = np.random.default_rng(seed)
    iy, ix = np.meshgrid(np.arange(segments + 1), np.arange(segments + 1), indexing="ij")
    v, u = iy / segments, ix / segments
    theta, phi = v * math.pi, u * 2 * math.pi
    d = np.stack([-np.cos(phi) * np.sin(theta), np.cos(theta), np.sin(phi) * np.sin(theta)], -1)
    r = np.ones(theta.shape)
    for o in range(octaves):
        f = 2.0 ** o
        k = rng.normal(size=(4, 3)) * f * 1.7
        ph = rng.uniform(0, 2 * math.pi, 4)
        amp = 0.22 / (f ** 0.9)
        for j in range(4):
            r = r + amp * np.sin(d @ k[j] + ph[j])
    r = np.maximum(r, 0.15)
    verts = (d * r[..., None]).reshape(-1, 3)
    # seam and poles: make duplicated grid vertices coincide exactly
    grid = verts.reshape(segments + 1, segments + 1, 3)
    grid[:, -1] = grid[:, 0]
    grid[0, :] = grid[0, 0]
    grid[-1, :] = grid[-1, 0]
    verts = grid.reshape(-1, 3)
    idx = np.arange((segments + 1) ** 2).reshape(segments + 1, segments + 1)
    a, b = idx[:-1, 1:], idx[:-1, :-1]
    c, dd = idx[1:, :-1], idx[1:, 1:]
    upper = np.stack([a, b, dd], -1)[1:].reshape(-1, 3)        # skipped on the first row (pole)
    lower = np.stack([b, c, dd], -1)[:-1].reshape(-1, 3)       # skipped on the last row
    faces = np.concatenate([upper, lower])
    return verts, faces


def dragon_class_scene(segments=660, seed=7):
    """BASELINE.md config 3 stand-in for the Stanford Dragon: ~870k triangles
    (segments=660 -> 869,880) of a bumpy closed surface, normalised like main.ts:268-279
    (scale = 1 / max(bounds.max), model at y = 0.5), on the 5x5 ground plane."""
    verts, faces = displaced_blob(segments, seed)
    verts = verts * (0.5 / float(np.abs(verts).max()))              # fits a unit cube
    verts[:, 1] -= verts[:, 1].min()                                # rest on the plane
    normals = _smooth_normals(verts, faces)
    pos = verts[faces]
    nrm = normals[faces]
    gp, gn, gm = _ground_plane()
    positions = np.concatenate([gp, pos])
    norms = np.concatenate([gn, nrm])
    mats = np.concatenate([gm, np.zeros(len(pos), np.int64)])
    return Scene(positions, norms, mats, [WHITE, RED], f"dragon-class-{len(positions)}")


def _tree_mesh(rng, foliage_segments=14):
    """One low-poly tree: a tapered 8-sided trunk + three stacked bumpy foliage blobs."""
    parts_v, parts_f = [], []
    n = 8
    ang = np.arange(n) * 2 * math.pi / n
    ring0 = np.stack([0.06 * np.cos(ang), np.zeros(n), 0.06 * np.sin(ang)], 1)
    ring1 = np.stack([0.035 * np.cos(ang), np.full(n, 0.45), 0.035 * np.sin(ang)], 1)
    tv = np.concatenate([ring0, ring1])
    tf = []
    for i in range(n):
        j = (i + 1) % n
        tf += [(i, n + i, j), (j, n + i, n + j)]
    parts_v.append(tv)
    parts_f.append(np.array(tf))
    off = len(tv)
    for k, (cy, rad) in enumerate([(0.55, 0.30), (0.80, 0.22), (1.0, 0.14)]):
        bv, bf = displaced_blob(foliage_segments, seed=int(rng.integers(1 << 30)), octaves=2)
        bv = bv * rad + np.array([0.0, cy, 0.0])
        parts_v.append(bv)
        parts_f.append(bf + off)
        off += len(bv)
    return np.concatenate(parts_v), np.concatenate(parts_f)


def forest_scene(instances=9000, seed=11, foliage_segments=14, area=40.0):
    """instanced-then-flattened trees on a ground plane, positions
    from a Halton (2,3) set; ~1.1k triangles per tree."""
    rng = np.random.default_rng(seed)
    variants = [_tree_mesh(rng, foliage_segments) for _ in range(8)]

    def halton(i, base):
        f, r = 1.0, 0.0
        while i > 0:
            f /= base
            r += f * (i % base)
            i //= base
        return r

    pos_list, nrm_list = [], []
    for i in range(instances):
        verts, faces = variants[i % len(variants)]
        normals = _smooth_normals(verts, faces)
        a = rng.uniform(0, 2 * math.pi)
        s = rng.uniform(0.7, 1.6)
        rot = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        t = np.array([(halton(i + 1, 2) - 0.5) * area, 0.0, (halton(i + 1, 3) - 0.5) * area])
        wv = (verts * s) @ rot.T + t
        wn = normals @ rot.T
        pos_list.append(wv[faces])
        nrm_list.append(wn[faces])
    g = flatten_mesh(plane_geometry(area * 1.2, area * 1.2),
                     compose_matrix(quaternion=quaternion_from_axis_angle((1.0, 0.0, 0.0), -math.pi / 2)), 0)
    positions = np.concatenate([g[0]] + pos_list)
    norms = np.concatenate([g[1]] + nrm_list)
    mats = np.zeros(len(positions), np.int64)
    green = dict(color=(0.35, 0.6, 0.3), roughness=1.0, metalness=0.0, specularColor=(1.0, 1.0, 1.0))
    mats[len(g[0]):] = 1
    sc = Scene(positions, norms, mats, [WHITE, green], f"forest-{len(positions)}")
    sc.camera.update(position=(0.0, 6.0, 24.0), target=(0.0, 0.5, 0.0), focalDistance=24.0)
    return sc
