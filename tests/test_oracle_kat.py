"""Known-answer tests that pin the CPU oracle (the reference has no tests or golden
vectors -- SURVEY.md 8c -- so every expected value here is derived by hand from the
WGSL / TypeScript source, or by an independent integer / float64 computation)."""
import math
import os

import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import layout, scenes

M32 = 0xFFFFFFFF


# ---------------------------------------------------------------- RNG (raytrace.wgsl:253-259)

def pcg_py(seed):
    """Independent integer implementation of the hash step."""
    seed = (seed * 747796405 + 2891336453) & M32
    word = (((seed >> ((seed >> 28) + 4)) ^ seed) * 277803737) & M32
    word = ((word >> 22) ^ word) & M32
    return seed, word


def test_pcg_sequence_is_integer_exact(orc):
    for start in (0, 1, 123456789, 0xDEADBEEF, M32):
        got, final = orc.rand_sequence(start, 64)
        s = start
        for i in range(64):
            s, w = pcg_py(s)
            want = np.float32(w) / np.float32(4294967296.0)    # f32(u32) RNE, divisor rounds to 2^32
            assert got[i] == want, (start, i)
        assert final == s


def _state_for_output(word):
    """Invert the output permutation: which state (after the LCG step) hashes to `word`?"""
    w = word ^ (word >> 22)                                   # x ^= x >> 22 is an involution on 32 bits
    w = (w * pow(277803737, -1, 1 << 32)) & M32
    shift = (w >> 28) + 4                                     # the top 4 bits survive the xorshift
    s = w
    for _ in range(8):
        s = w ^ (s >> shift)
    assert (((s >> ((s >> 28) + 4)) ^ s) & M32) == w
    return s


def test_rand_endpoints(orc):
    """rand() can return exactly 0.0 and -- because 4294967295.0 rounds to 2^32 and
    f32(0xFFFFFFFF) rounds up to 2^32 -- exactly 1.0."""
    inv = pow(747796405, -1, 1 << 32)
    for word, expect in ((0, 0.0), (M32, 1.0), (0xFFFFFF80, 1.0), (0xFFFFFF7F, np.float32(0xFFFFFF00) / np.float32(2.0 ** 32))):
        state = _state_for_output(word)
        seed = ((state - 2891336453) * inv) & M32
        got, _ = orc.rand_sequence(seed, 1)
        assert got[0] == np.float32(expect), hex(word)


# ---------------------------------------------------------------- slab test (raytrace.wgsl:118-152)

BOX = ((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))


@pytest.mark.parametrize("o,d,expect", [
    ((0, 0, 5), (0, 0, -1), True),            # straight on; x and y axes take the parallel branch
    ((0, 0, 5), (0, 0, 1), False),            # pointing away: tmax < 0
    ((0, 0, 0), (0, 0, 1), True),             # origin inside
    ((2, 0, 5), (0, 0, -1), False),           # parallel axis, origin outside the x slab
    ((1, 0, 5), (0, 0, -1), True),            # parallel axis exactly on the face: inside (o > max is false)
    ((0, 0, 5), (1e-7, 0, -1), True),         # |d.x| < EPSILON counts as parallel
    ((3, 0, 5), (1e-7, 0, -1), False),
    ((0, 0, 5), (0.6, 0, -0.8), False),       # exits the x slab before entering z
    ((0, 0, 5), (0.2, 0, -0.9797959), True),
    ((-5, 1, 0), (1, 0, 0), True),            # grazing along the top face (parallel y, o.y == max)
    ((-5, 1.0000001, 0), (1, 0, 0), False),
    ((-2, -2, -2), (1, 1, 1), True),          # through both corners: tmin == tmax on every axis
])
def test_ray_aabb(orc, o, d, expect):
    assert orc.ray_aabb(o, d, *BOX) is expect


def test_ray_aabb_flat_box(orc):
    # zero-thickness box (axis-aligned triangle): t1 == t2 on that axis, hit when inside the footprint
    assert orc.ray_aabb((0.2, 3, 0.1), (0, -1, 0), (-1, 0, -1), (1, 0, 1))
    assert not orc.ray_aabb((1.2, 3, 0.1), (0, -1, 0), (-1, 0, -1), (1, 0, 1))
    assert orc.ray_aabb((0.2, 3, 0.1), (0.1, -0.99, 0.05), (-1, 0, -1), (1, 0, 1))


# ---------------------------------------------------------------- triangle (raytrace.wgsl:78-116)

def _tri(a, b, c, normals=None, material=3):
    t = np.zeros(1, layout.TRIANGLE)
    t["aPosition"], t["bPosition"], t["cPosition"] = a, b, c
    n = normals or ((0, 0, 1),) * 3
    t["aNormal"], t["bNormal"], t["cNormal"] = n
    t["materialIndex"] = material
    return t


def test_moller_trumbore_cases(orc):
    tri = _tri((0, 0, 0), (2, 0, 0), (0, 2, 0))
    h = orc.ray_triangle((0.5, 0.5, 3), (0, 0, -1), tri)
    assert h[0] == 1 and h[1] == 3.0 and tuple(h[2:5]) == (0.5, 0.5, 0.0) and tuple(h[5:8]) == (0, 0, 1) and h[8] == 3
    # two-sided: from behind, same normal (not flipped)
    h = orc.ray_triangle((0.5, 0.5, -3), (0, 0, 1), tri)
    assert h[0] == 1 and h[1] == 3.0 and tuple(h[5:8]) == (0, 0, 1)
    # on an edge (u = 0) and on a vertex: inclusive bounds u >= 0, v >= 0, u + v <= 1
    assert orc.ray_triangle((0.0, 1.0, 1), (0, 0, -1), tri)[0] == 1
    assert orc.ray_triangle((0.0, 0.0, 1), (0, 0, -1), tri)[0] == 1
    assert orc.ray_triangle((1.0, 1.0, 1), (0, 0, -1), tri)[0] == 1      # hypotenuse: u + v == 1
    assert orc.ray_triangle((1.01, 1.0, 1), (0, 0, -1), tri)[0] == 0
    assert orc.ray_triangle((-0.01, 1.0, 1), (0, 0, -1), tri)[0] == 0
    # t <= EPSILON rejected (self-hit guard), t slightly larger accepted
    assert orc.ray_triangle((0.5, 0.5, 1e-6), (0, 0, -1), tri)[0] == 0
    assert orc.ray_triangle((0.5, 0.5, 2e-6), (0, 0, -1), tri)[0] == 1
    # parallel ray: |a| < EPSILON
    assert orc.ray_triangle((0.5, 0.5, 1), (1, 0, 0), tri)[0] == 0
    # miss keeps t = INF (1e20) and the triangle's material index
    m = orc.ray_triangle((5, 5, 1), (0, 0, -1), tri)
    assert m[0] == 0 and m[1] == np.float32(1e20) and m[8] == 3


def test_interpolated_normal_is_normalised_not_flipped(orc):
    tri = _tri((0, 0, 0), (1, 0, 0), (0, 1, 0), normals=((0, 0, 2), (0, 1, 0), (1, 0, 0)))
    h = orc.ray_triangle((0.25, 0.25, 1), (0, 0, -1), tri)
    w, u, v = 0.5, 0.25, 0.25
    n = np.array([v * 1, u * 1, w * 2])
    n /= np.linalg.norm(n)
    assert np.allclose(h[5:8], n, rtol=1e-6)


# ---------------------------------------------------------------- traversal (raytrace.wgsl:154-203)

def _two_leaf_scene(orc, tri_a, tri_b):
    tris = np.zeros(2, layout.TRIANGLE)       # (np.concatenate would repack the padded dtype)
    tris[0], tris[1] = tri_a[0], tri_b[0]
    nodes = np.zeros(3, layout.BVH_NODE)
    pts = np.concatenate([tris["aPosition"], tris["bPosition"], tris["cPosition"]])
    nodes[0]["min"], nodes[0]["max"] = pts.min(0), pts.max(0)
    nodes[0]["isLeaf"], nodes[0]["left"], nodes[0]["right"], nodes[0]["triangleIndex"] = 0, 1, 2, -1
    for i in (0, 1):
        p = np.stack([tris[i]["aPosition"], tris[i]["bPosition"], tris[i]["cPosition"]])
        nodes[1 + i]["min"], nodes[1 + i]["max"] = p.min(0), p.max(0)
        nodes[1 + i]["isLeaf"], nodes[1 + i]["left"], nodes[1 + i]["right"], nodes[1 + i]["triangleIndex"] = 1, -1, -1, i
    mats = layout.pack_materials([scenes.WHITE] * 8)
    return orc.OracleScene(tris, mats, nodes), tris, mats, nodes


def test_tie_break_first_visited_wins_and_right_child_is_visited_first(orc):
    """Two coincident triangles with different materials: equal t, strict '<'
    (raytrace.wgsl:180) keeps the first visited; left is pushed first so the RIGHT leaf is
    popped first (raytrace.wgsl:184-198)."""
    a = _tri((-1, -1, 0), (1, -1, 0), (0, 1, 0), material=1)
    b = _tri((-1, -1, 0), (1, -1, 0), (0, 1, 0), material=2)
    sc, *_ = _two_leaf_scene(orc, a, b)
    h, cnt = orc.ray_scene(sc, (0, 0, 2), (0, 0, -1))
    assert h[0] == 1 and h[1] == 2.0 and h[8] == 2          # the right leaf's triangle
    assert cnt["box_tests"] == 3 and cnt["tri_tests"] == 2
    sc, *_ = _two_leaf_scene(orc, b, a)
    assert orc.ray_scene(sc, (0, 0, 2), (0, 0, -1))[0][8] == 1


def test_closest_hit_and_counters(orc):
    near = _tri((-1, -1, 1), (1, -1, 1), (0, 1, 1), material=4)
    far = _tri((-1, -1, -1), (1, -1, -1), (0, 1, -1), material=5)
    sc, *_ = _two_leaf_scene(orc, far, near)
    h, cnt = orc.ray_scene(sc, (0, 0, 3), (0, 0, -1))
    assert h[8] == 4 and h[1] == 2.0
    # a ray that misses the root box costs exactly one box test and no pops
    h, cnt = orc.ray_scene(sc, (5, 5, 3), (0, 0, -1))
    assert h[0] == 0 and cnt["box_tests"] == 1 and cnt["tri_tests"] == 0 and cnt["rays"] == 1


def test_empty_bvh_always_misses(orc):
    sc = orc.OracleScene(np.zeros(2, layout.TRIANGLE), layout.pack_materials([scenes.WHITE]), None)
    h, cnt = orc.ray_scene(sc, (0, 0, 3), (0, 0, -1))
    assert h[0] == 0 and cnt["box_tests"] == 0


# ---------------------------------------------------------------- camera (raytrace.wgsl:217-250)

class _Cam:
    def __init__(self, position, direction, fov=45.0):
        self.camera = dict(position=position, fov=fov, focalDistance=1.0, aperture=0.0)
        self._d = direction

    def camera_direction(self):
        return self._d


def _camera_ref(pos, direction, fov, aspect, uvx, uvy):
    """Independent float64 evaluation of cameraToRay, including its quirks."""
    t = math.tan(math.radians(fov) / 2)
    r = aspect * t
    u = -r + 2 * r * uvx
    v = -t + 2 * t * uvy
    w = -np.array(direction, float)
    w /= np.linalg.norm(w)
    up = np.array([0.0, 1.0, 0.0])
    if abs(w @ up) > 0.99999:
        up = np.array([0.0, 0.0, 1.0])
    ud = np.cross(up, w)
    ud /= np.linalg.norm(ud)
    vd = np.cross(w, ud)
    d = ud * u + vd * v - w * aspect              # forward term is -w * aspect (raytrace.wgsl:238)
    return d / np.linalg.norm(d)


@pytest.mark.parametrize("direction", [(0, -1, -4), (0, -1, 0), (0, 1, 0), (1, 0, 0), (0.3, -0.2, -0.9)])
@pytest.mark.parametrize("aspect", [1.0, 16 / 9])
def test_camera_rays(orc, direction, aspect):
    d = np.array(direction, float)
    d /= np.linalg.norm(d)
    cam = _Cam((0.0, 1.0, 4.0), tuple(d))
    u = pc.rt_uniforms(cam, 160, 90, aspect=aspect)
    for uvx, uvy in ((0, 0), (1, 0), (0, 1), (1, 1), (0.5, 0.5), (0.25, 0.75)):
        got = orc.camera_ray(u.tobytes(), uvx, uvy)
        assert tuple(got[:3]) == (0.0, 1.0, 4.0)
        want = _camera_ref((0, 1, 4), d, 45.0, aspect, uvx, uvy)
        assert np.allclose(got[3:], want, atol=2e-6), (uvx, uvy, got[3:], want)
    # the centre ray is the camera direction whatever the aspect (u = v = 0)
    assert np.allclose(orc.camera_ray(u.tobytes(), 0.5, 0.5)[3:], d, atol=2e-6)


def test_camera_effective_fov_is_narrowed_by_aspect(orc):
    """Because the forward term is -w*aspect, the vertical half-angle is
    atan(tan(fov/2)/aspect), not fov/2."""
    cam = _Cam((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), fov=90.0)
    u = pc.rt_uniforms(cam, 200, 100, aspect=2.0)
    top = orc.camera_ray(u.tobytes(), 0.5, 1.0)[3:]
    assert math.isclose(math.degrees(math.atan2(top[1], -top[2])), math.degrees(math.atan(1.0 / 2.0)), abs_tol=1e-4)


# ---------------------------------------------------------------- environment lookup

def test_equirect_uv(orc, demo):
    u = pc.rt_uniforms(demo, 8, 8)
    cases = {(0, 0, 1): (0.5, 0.5), (1, 0, 0): (0.75, 0.5), (-1, 0, 0): (0.25, 0.5), (0, 0, -1): (1.0, 0.5),
             (0, 1, 0): (0.5, 0.0), (0, -1, 0): (0.5, 1.0)}
    for d, uv in cases.items():
        got = orc.env_uv(u.tobytes(), d)
        assert np.allclose(got, uv, atol=1e-6), (d, got)
    # a non-unit direction (mix() output is not re-normalised, raytrace.wgsl:391) is clamped, not normalised
    assert np.allclose(orc.env_uv(u.tobytes(), (0, 3, 0)), (0.5, 0.0), atol=1e-6)
    # rotation about Y by +90 degrees maps +X onto -Z... measured in uv: u moves by +0.25
    ur = pc.rt_uniforms(demo, 8, 8, rotation=math.pi / 2)
    assert np.allclose(orc.env_uv(ur.tobytes(), (0, 0, 1)), (0.25, 0.5), atol=1e-6)


def test_bilinear_clamp_and_repeat(orc):
    env = np.zeros((512, 1024, 4), np.float32)
    env[0, 0, :3] = 8.0
    env[0, 1, :3] = 4.0
    env[1, 0, :3] = 2.0
    env[0, 1023, :3] = 100.0
    sc = orc.OracleScene(np.zeros(2, layout.TRIANGLE), layout.pack_materials([scenes.WHITE]), None, env)
    # texel centres return the texel
    assert tuple(orc.sample_env(sc, 0.5 / 1024, 0.5 / 512)) == (8.0, 8.0, 8.0)
    # half-way between texel (0,0) and (1,0)
    assert tuple(orc.sample_env(sc, 1.0 / 1024, 0.5 / 512)) == (6.0, 6.0, 6.0)
    # centre of the 2x2 block: (8 + 4 + 2 + 0) / 4
    assert tuple(orc.sample_env(sc, 1.0 / 1024, 1.0 / 512)) == (3.5, 3.5, 3.5)
    # clamp-to-edge: the seam u = 0 does NOT wrap to the last column (renderer.ts:77-80)
    assert tuple(orc.sample_env(sc, 0.0, 0.5 / 512)) == (8.0, 8.0, 8.0)
    assert tuple(orc.sample_env(sc, -3.0, 0.5 / 512)) == (8.0, 8.0, 8.0)
    # the fullscreen sampler repeats (fullscreen.ts:49-57): u = 0 blends column 0 with column W-1
    tex = np.zeros((4, 4, 4), np.float32)
    tex[0, 0] = 8.0
    tex[0, 3] = 100.0
    assert orc.sample_repeat(tex, 0.0, 0.125)[0] == 54.0
    assert orc.sample_repeat(tex, 1.0, 0.125)[0] == 54.0
    assert orc.sample_repeat(tex, 1.125, 1.125)[0] == 8.0


# ---------------------------------------------------------------- accumulate / fullscreen

def test_accumulate_running_mean_law(orc):
    """First sampled frame carries frame = 2 (renderer.ts:369-377), so after n frames of a
    constant colour c the buffer holds c * n / (n + 1)."""
    w = h = 4
    c = np.full((h, w, 4), 3.0, np.float32)
    acc = np.zeros_like(c)
    for n, frame in enumerate((2, 3, 4), start=1):
        acc = orc.accumulate(pc.acc_uniforms(w, h, frame).tobytes(), w, h, c, acc)
        assert np.allclose(acc[..., :3], 3.0 * n / (n + 1), rtol=1e-6)
        assert (acc[..., 3] == 1.0).all()
    # disabled -> weight 1 -> the frame passes through
    out = orc.accumulate(pc.acc_uniforms(w, h, 9, enabled=0).tobytes(), w, h, c, acc)
    assert (out[..., :3] == 3.0).all()
    # frame 0 keeps weight 1 too (accumulate.wgsl:21-24)
    out = orc.accumulate(pc.acc_uniforms(w, h, 0).tobytes(), w, h, c, acc)
    assert (out[..., :3] == 3.0).all()
    # only texels inside `resolution` are touched
    out = orc.accumulate(pc.acc_uniforms(2, 3, 2).tobytes(), w, h, c, np.zeros_like(c))
    assert (out[:3, :2, :3] == 1.5).all() and not out[3:].any() and not out[:, 2:].any()


def test_fp16_storage_rounding(orc):
    w = h = 2
    c = np.full((h, w, 4), 0.1, np.float32)
    acc = orc.accumulate(pc.acc_uniforms(w, h, 3).tobytes(), w, h, c, np.zeros_like(c), store_f16=True)
    want = np.float32(np.float16(np.float32(0.1) * np.float32(1 / 3) + 0))   # prev 0: 0*(1-w) + c*w
    assert acc[0, 0, 0] == np.float32(np.float16(np.float32(0) * (1 - np.float32(1) / 3) + np.float32(0.1) * (np.float32(1) / 3)))
    assert abs(acc[0, 0, 0] - want) <= 2.0 ** -14


def test_denoise_identity_and_step_edge(orc):
    w, h = 32, 24
    flat = np.full((h, w, 4), 0.37, np.float32)
    flat[..., 3] = 1.0
    f = pc.fs_uniforms(w, h, 1.0, denoise=1, tonemapping=0)
    out, _ = orc.fullscreen(f.tobytes(), flat)
    assert np.allclose(out[..., :3], 0.37, rtol=2e-6)
    # step edge: far from the edge nothing changes; everything stays within [lo, hi];
    # the bilateral threshold (0.08) keeps the edge sharp (a 0.9 step is >> threshold)
    img = np.full((h, w, 4), 0.05, np.float32)
    img[:, w // 2:, :3] = 0.95
    img[..., 3] = 1.0
    out, _ = orc.fullscreen(f.tobytes(), img)
    view = out[::-1]                       # canvas row 0 is the top, texture row 0 the bottom
    assert np.allclose(view[:, 8, :3], 0.05, rtol=1e-5) and np.allclose(view[:, 24, :3], 0.95, rtol=1e-5)
    assert view[..., :3].min() >= 0.05 - 1e-6 and view[..., :3].max() <= 0.95 + 1e-6
    assert np.allclose(view[10, w // 2 - 1, :3], 0.05, atol=1e-3) and np.allclose(view[10, w // 2, :3], 0.95, atol=1e-3)


def test_fullscreen_orientation_and_tonemaps(orc):
    """Texture row 0 (camera down) is drawn at the BOTTOM of the canvas; Reinhard is
    x / (1 + x); ACES maps 0 -> 0 and saturates to 1."""
    w, h = 8, 6
    img = np.zeros((h, w, 4), np.float32)
    img[0, :, :3] = 1.0                    # bottom texture row white
    img[..., 3] = 1.0
    out, o8 = orc.fullscreen(pc.fs_uniforms(w, h, 1.0, denoise=0, tonemapping=2).tobytes(), img)
    # (the bilinear weights at texel centres are 1 - eps / eps in fp32, hence allclose)
    assert np.allclose(out[h - 1, :, :3], 0.5, rtol=1e-6) and (out[0, :, :3] == 0.0).all()
    assert np.isin(o8[h - 1, :, :3], (127, 128)).all() and (o8[..., 3] == 255).all()
    out, _ = orc.fullscreen(pc.fs_uniforms(w, h, 1.0, denoise=0, tonemapping=1).tobytes(), img)
    assert (out[0, :, :3] == 0.0).all()
    big = img.copy()
    big[..., :3] = 1000.0
    out, _ = orc.fullscreen(pc.fs_uniforms(w, h, 1.0, denoise=0, tonemapping=1).tobytes(), big)
    assert (out[..., :3] == 1.0).all()
    # ACES against an independent float64 evaluation of the Hill fit + gamma
    x = np.array([0.18, 0.5, 2.0])
    m1 = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
    m2 = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
    v = m1 @ x
    r = (v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081)
    want = np.clip(m2 @ r, 0, 1) ** (1 / 2.2)
    img2 = np.zeros((2, 2, 4), np.float32)
    img2[..., :3] = x
    out, _ = orc.fullscreen(pc.fs_uniforms(2, 2, 1.0, denoise=0, tonemapping=1).tobytes(), img2)
    assert np.allclose(out[0, 0, :3], want, rtol=2e-5)


# ---------------------------------------------------------------- whole-frame invariants

def test_frame_counters_and_black_frame(orc, demo, env):
    sc = pc.oracle_scene(orc, demo, env)
    w = h = 24
    img, cnt = orc.raytrace(sc, pc.rt_uniforms(demo, w, h, bounces=0).tobytes(), w, h)
    assert cnt["rays"] == 0 and cnt["pixels"] == w * h and not img[..., :3].any() and (img[..., 3] == 1).all()
    img, cnt = orc.raytrace(sc, pc.rt_uniforms(demo, w, h, bounces=8).tobytes(), w, h)
    assert cnt["rays"] == cnt["hits"] + cnt["misses"]
    assert cnt["pixels"] <= cnt["rays"] <= 8 * cnt["pixels"]
    assert (cnt["box_tests"] - cnt["rays"]) % 2 == 0         # N_box = 1 + 2 * internal pops per ray
    # samplesPerFrame = 2 is the mean of two consecutive sample paths drawn from one seed stream
    img2, cnt2 = orc.raytrace(sc, pc.rt_uniforms(demo, w, h, bounces=2, spf=2).tobytes(), w, h)
    assert cnt2["pixels"] == w * h and cnt2["rays"] >= 2 * w * h


def test_tile_split_matches_whole_image(orc, demo, env):
    sc = pc.oracle_scene(orc, demo, env)
    w, h = 40, 37
    u = pc.rt_uniforms(demo, w, h, bounces=3)
    whole, cnt = orc.raytrace(sc, u.tobytes(), w, h)
    for nranks, block in ((2, 8), (3, 5), (8, 8)):
        out = np.zeros_like(whole)
        rays = 0
        for rank in range(nranks):
            part, c = orc.raytrace(sc, u.tobytes(), w, h, rank, nranks, block)
            from mi3pt_host import tiles
            rows = tiles.local_rows_of(h, rank, nranks, block)
            assert part.shape[0] == len(rows) == orc.tile_local_rows(h, rank, nranks, block)
            out[rows] = part
            rays += c["rays"]
        assert pc.same_bits(out, whole) and rays == cnt["rays"]


def test_oracle_threads_follow_the_cpu_share(orc, tmp_path, monkeypatch):
    """The oracle's passes run on the box's CPU share (cgroup quota), not on every processor OpenMP sees: 256 threads on a
    16-core share ran it at a tenth of its speed (round-4 verdict, weak #1)."""
    n = orc.default_threads()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    q = orc.cpu_quota()
    assert q is None or n <= q
    orc.set_num_threads(n)
    assert orc.max_threads() == n
    # the two file formats (read through the same code with the paths redirected)
    import builtins
    real_open = builtins.open
    files = {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
            if path in files:
                f = tmp_path / "f"
                f.write_text(files[path])
                return real_open(f, *a, **k)
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert orc.cpu_quota() == 16
    files.clear()
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert orc.cpu_quota() is None
    files.clear()
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "800000\n"
    files["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000\n"
    assert orc.cpu_quota() == 8
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"
    assert orc.cpu_quota() is None
