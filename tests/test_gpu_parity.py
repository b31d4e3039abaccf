"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

The bar (BASELINE.json north_star): per-pixel agreement within 1e-4 relative fp32 at the
fixed RNG seed.  Because the product and the oracle implement the same pinned
arithmetic, the tests ask for more: BIT-identical images and identical ray / box-test /
triangle-test counters; the 1e-4 check is kept as the documented tolerance.
"""
import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi, layout, scenes

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4      # north_star tolerance, relative, per pixel value


# ---------------------------------------------------------------- math

MATH_CASES = [
    (0, "sin", lambda r: r.uniform(-10, 10, 200000), None),
    (1, "cos", lambda r: r.uniform(-10, 10, 200000), None),
    (0, "sin-wide", lambda r: r.uniform(-3000, 3000, 100000), None),
    (2, "tan", lambda r: r.uniform(-1.5, 1.5, 100000), None),
    (3, "log", lambda r: np.concatenate([r.uniform(0, 1, 100000), 2.0 ** -r.uniform(0, 149, 50000),
                                         r.uniform(0, 1e9, 20000), [0.0, -1.0, np.inf, np.nan, 1.0]]), None),
    (4, "exp", lambda r: np.concatenate([r.uniform(-110, 92, 200000), [-np.inf, np.inf, np.nan, 0.0]]), None),
    (5, "atan2", lambda r: r.normal(size=200000), lambda r: r.normal(size=200000)),
    (6, "asin", lambda r: np.concatenate([r.uniform(-1, 1, 200000), [-1, 1, 0, 1.5, 1e-5, -1e-5]]), None),
    (7, "pow", lambda r: np.concatenate([r.uniform(0, 1, 100000), [0.0, 1.0]]),
     lambda r: np.full(100002, 1 / 2.2)),
    (8, "f16", lambda r: np.concatenate([r.normal(size=100000) * 10.0 ** r.uniform(-9, 6, 100000),
                                         [0, 65504, 65519.9, 65520, 1e-8, np.inf, -np.inf, 5.96e-8, 2.98e-8]]),
     None),
    (9, "sqrt", lambda r: r.uniform(0, 1e6, 100000), None),
    (10, "div", lambda r: r.normal(size=200000) * 100, lambda r: r.normal(size=200000)),
]


@pytest.mark.parametrize("fn,name,gen_a,gen_b", MATH_CASES, ids=[c[1] for c in MATH_CASES])
def test_math_bit_exact(gpu_ctx, orc, fn, name, gen_a, gen_b):
    rng = np.random.default_rng(1234 + fn)
    a = gen_a(rng).astype(np.float32)
    b = gen_b(rng).astype(np.float32) if gen_b else None
    got = gpu_ctx.debug_math(fn, a, b)
    want = orc.math_fn(fn, a, b)
    assert pc.same_bits(got, want), f"{name}: {pc.describe_diff(got, want)}"


def _edge_values(rng, n):
    """Magnitudes over the whole exponent range, with the guard boundaries of the reduced forms,
    zeros, subnormals and non-finite values mixed in."""
    v = rng.choice([-1.0, 1.0], n) * rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-149, 128, n)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, 2.0 ** -64, 2.0 ** 64, 2.0 ** -80, 2.0 ** -40,
                        2.0 ** 40, 2.0 ** -126, 2.0 ** -149, 3.4e38, np.nextafter(np.float32(2.0 ** -64), np.float32(0)),
                        np.nextafter(np.float32(2.0 ** 64), np.float32(np.inf))])
    v[:: max(n // 4096, 1)][: 4096] = rng.choice(special, len(v[:: max(n // 4096, 1)][: 4096]))
    return np.concatenate([v, special]).astype(np.float32)


def test_log_for_rand_values_equals_the_oracles_log(gpu_ctx, orc):
    """ptm::log1_unit (pt_devmath.h: randNormal's log without the tests no value of rand() needs) against the oracle's log
    (fn 3) on values of the form f32(r) / 2^32 -- every exponent, the ends 0, 2^-32 and 1.0, the rounding up to 1.0 -- and
    against the device's own log1 (the exhaustive comparison is profiles/log_unit_proof.hip; this is the regression test)."""
    rng = np.random.default_rng(4242)
    r = np.concatenate([rng.integers(0, 2 ** 32, 600000, dtype=np.uint64),
                        (rng.integers(1, 2 ** 24, 200000, dtype=np.uint64) << rng.integers(0, 9, 200000).astype(np.uint64)),
                        rng.integers(0, 2 ** 12, 100000, dtype=np.uint64),
                        np.array([0, 1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 128, 2 ** 32 - 129, 2 ** 31, 2 ** 31 - 1, 3037000500], dtype=np.uint64)])
    x = (r.astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)
    assert x.min() == 0.0 and x.max() == 1.0 and x[x > 0].min() == np.float32(2.0 ** -32)
    got = gpu_ctx.debug_math(16, x)
    with np.errstate(all="ignore"):
        assert pc.same_bits(got, orc.math_fn(3, x, None)), pc.describe_diff(got, orc.math_fn(3, x, None))
    assert pc.same_bits(got, gpu_ctx.debug_math(3, x))


def test_reduced_sqrt_rcp_normalize_equal_the_ieee_operations(gpu_ctx):
    """ptm::sqrt_exact / rcp_exact / the normalize built on them (pt_devmath.h) against numpy's
    correctly rounded float32 sqrt and division, over all exponents and the guard boundaries
    (the exhaustive checks are profiles/div_proof.hip; this is the regression test)."""
    rng = np.random.default_rng(77)
    with np.errstate(all="ignore"):
        x = _edge_values(rng, 400000)
        assert pc.same_bits(gpu_ctx.debug_math(11, np.abs(x)), np.sqrt(np.abs(x)))
        assert pc.same_bits(gpu_ctx.debug_math(11, x[:4096]), np.sqrt(x[:4096]))            # negative -> NaN
        assert pc.same_bits(gpu_ctx.debug_math(12, x), np.float32(1.0) / x)
        for scale in (1.0, 2.0 ** -38, 2.0 ** 39, 2.0 ** -70):
            a = (rng.normal(size=300000) * scale).astype(np.float32)
            b = (rng.normal(size=300000) * scale * 10.0 ** rng.uniform(-30, 2, 300000)).astype(np.float32)
            a[::7] = 0.0
            b[::11] = 0.0
            b[::13] = a[::13]                                    # z = a - b = 0
            c = a - b
            length = np.sqrt((a * a + b * b) + c * c)
            for fn, comp in ((13, a), (14, b), (15, c)):
                assert pc.same_bits(gpu_ctx.debug_math(fn, a, b), comp / length), (scale, fn)
        a, b = _edge_values(rng, 100000), _edge_values(rng, 100000)[::-1].copy()
        c = a - b
        length = np.sqrt((a * a + b * b) + c * c)
        for fn, comp in ((13, a), (14, b), (15, c)):
            assert pc.same_bits(gpu_ctx.debug_math(fn, a, b), comp / length), fn


# ---------------------------------------------------------------- intersection probes

def _random_rays(rng, n, origin_scale=3.0):
    o = rng.normal(size=(n, 3)) * origin_scale
    t = rng.normal(size=(n, 3)) * 0.8 + np.array([0.0, 0.4, 0.0])
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("variant", [1, 2, 4])
def test_ray_scene_intersect_matches_oracle(gpu_ctx, orc, demo, env, variant):
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_kernel_variant(variant)
    rng = np.random.default_rng(5)
    rays = _random_rays(rng, 4000)
    # axis-parallel and grazing rays exercise the |d| < EPSILON branch of the slab test
    special = np.array([[0, 5, 0.5, 0, -1, 0], [0, 0.4, 5, 0, 0, -1], [-5, 0.4, 0.5, 1, 0, 0],
                        [0.4, 5, 0.5, 0, -1, 0], [0.4, 5, 0.9, 0, -1, 0], [0, 0, 0, 0, 1, 0],
                        [2.5, 1, 2.5, 0, -1, 0], [0, 1e-7, 0, 1, 0, 0]], np.float32)
    # directions whose significand is all ones / origins outside the guarded range force the
    # exact-division fallback of the prepared-reciprocal slab test (RayPre::slow)
    ones = np.frombuffer(np.array([0x3f7fffff, 0x3effffff, 0xbf7fffff, 0x3f7ffff3], np.uint32).tobytes(), np.float32)
    guard = np.array([[0.3, 2.0, 0.2, ones[1], -ones[0], ones[1] * 0.5], [0, 3, 0.5, 0.1, ones[2], 0.05],
                      [1e-30, 3, 0.5, 0.01, -1, 0.02], [0.1, 3e20, 0.5, 0, -1, 0], [0.2, 2, 0.1, ones[3] * 0.1, -0.9, 0.1]],
                     np.float32)
    rays = np.concatenate([rays, special, guard])
    got = ctx.debug_intersect(rays)
    osc = pc.oracle_scene(orc, demo, env)
    for i, r in enumerate(rays):
        want, cnt = orc.ray_scene(osc, r[:3], r[3:])
        assert pc.same_bits(got[i, :9], want), f"ray {i} {r}: gpu {got[i]} oracle {want}"
        assert (int(got[i, 9]), int(got[i, 10]), int(got[i, 11])) == \
            (cnt["box_tests"], cnt["tri_tests"], cnt["stack_overflows"]), f"ray {i} counters"
    ctx.set_kernel_variant(0)


def _chain_scene(depth):
    """A degenerate right-deep BVH (hand-made, not from the builder): a chain of `depth`
    internal nodes whose LEFT children are leaves and whose right child continues the
    chain.  Every box is hit, so the stack grows by one per level: with depth >= 64 the
    walk must abort at raytrace.wgsl:167-171 with best-so-far."""
    ntri = depth + 1
    pos = np.zeros((ntri, 3, 3))
    for i in range(ntri):
        z = -1.0 - i        # triangles stacked along -z, all hit by the ray (0,0,5)->-z
        pos[i] = [[-1, -1, z], [1, -1, z], [0, 1, z]]
    nrm = np.tile(np.array([0.0, 0.0, 1.0]), (ntri, 3, 1))
    tris = layout.pack_triangles(pos, nrm, np.zeros(ntri, int))
    nodes = np.zeros(2 * ntri - 1, layout.BVH_NODE)
    # node 2k = internal k (k < depth), node 2k+1 = leaf k, last node = leaf `depth`
    # breadth-first-like order with child > parent
    def box(lo_tri, hi_tri):
        p = pos[lo_tri:hi_tri + 1].reshape(-1, 3)
        return p.min(0), p.max(0)
    idx = 0
    for k in range(depth):
        mn, mx = box(k, depth)
        nodes[idx]["min"], nodes[idx]["max"] = mn, mx
        nodes[idx]["isLeaf"] = 0
        # children: right (idx+2) = the rest of the chain is pushed LAST and popped first,
        # so leaves pile up on the stack
        nodes[idx]["left"], nodes[idx]["right"] = idx + 1, idx + 2
        nodes[idx]["triangleIndex"] = -1
        mn, mx = box(k, k)
        nodes[idx + 1]["min"], nodes[idx + 1]["max"] = mn, mx
        nodes[idx + 1]["isLeaf"], nodes[idx + 1]["left"], nodes[idx + 1]["right"] = 1, -1, -1
        nodes[idx + 1]["triangleIndex"] = k
        idx += 2
    mn, mx = box(depth, depth)
    nodes[idx]["min"], nodes[idx]["max"] = mn, mx
    nodes[idx]["isLeaf"], nodes[idx]["left"], nodes[idx]["right"], nodes[idx]["triangleIndex"] = 1, -1, -1, depth
    mats = layout.pack_materials([scenes.WHITE])
    return tris, mats, nodes


@pytest.mark.parametrize("variant", [1, 2, 4])
@pytest.mark.parametrize("depth", [10, 62, 63, 64, 70])
def test_stack_overflow_abort_matches_oracle(gpu_ctx, orc, variant, depth):
    tris, mats, nodes = _chain_scene(depth)
    ctx = gpu_ctx
    ctx.upload_bvh(nodes)
    ctx.upload_triangles(tris)
    ctx.upload_materials(mats)
    ctx.set_kernel_variant(variant)
    rays = np.array([[0, 0, 5, 0, 0, -1], [0.2, -0.3, 5, 0, 0, -1], [0, 0, -200, 0, 0, 1]], np.float32)
    got = ctx.debug_intersect(rays)
    osc = orc.OracleScene(tris, mats, nodes)
    overflowed = 0
    for i, r in enumerate(rays):
        want, cnt = orc.ray_scene(osc, r[:3], r[3:])
        assert pc.same_bits(got[i, :9], want), f"depth {depth} ray {i}: gpu {got[i]} oracle {want}"
        assert int(got[i, 11]) == cnt["stack_overflows"]
        assert int(got[i, 9]) == cnt["box_tests"] and int(got[i, 10]) == cnt["tri_tests"]
        overflowed += cnt["stack_overflows"]
    if depth >= 64:
        assert overflowed > 0, "the chain is deep enough that the reference walk must abort"
    ctx.set_kernel_variant(0)


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 7])
def test_stack_overflow_inside_a_frame(gpu_ctx, orc, env, variant):
    """The 64-entry abort must also come out right inside the full raytrace pass of every
    kernel variant (the persistent kernels keep the stack pointer across steps)."""
    tris, mats, nodes = _chain_scene(70)
    ctx = gpu_ctx
    ctx.upload_bvh(nodes)
    ctx.upload_triangles(tris)
    ctx.upload_materials(mats)
    ctx.upload_environment(env)
    pc.set_variant_or_skip(ctx, variant)
    ctx.set_tile(0, 1, 8)
    w = h = 24
    ctx.resize(w, h)
    ctx.reset_counters()

    class Cam:
        camera = dict(position=(0.0, 0.0, 5.0), fov=30.0, focalDistance=1.0, aperture=0.0)

        @staticmethod
        def camera_direction():
            return (0.0, 0.0, -1.0)

    u = pc.rt_uniforms(Cam, w, h, frame=2, bounces=3)
    pc.gpu_frame(ctx, u)
    got = ctx.read_texture(capi.TEX_OUTPUT)
    cnt = ctx.counters()
    want, ocnt = orc.raytrace(orc.OracleScene(tris, mats, nodes, env), u.tobytes(), w, h)
    assert ocnt["stack_overflows"] > 0
    assert pc.same_bits(got, want), pc.describe_diff(got, want)
    for k in ("rays", "box_tests", "tri_tests", "hits", "misses", "stack_overflows", "pixels"):
        assert cnt[k] == ocnt[k], f"counter {k}: gpu {cnt[k]} oracle {ocnt[k]}"
    ctx.set_kernel_variant(0)


def _coincident_sheets_scene(copies=12, segments=3):
    """`copies` coincident subdivided quads, each with its own emissive material: every hit is a
    `copies`-way tie in t, so the colour of a pixel says which leaf the walk visited first."""
    import math
    from mi3pt_host import scenes as S
    q = S.quaternion_from_axis_angle((1.0, 0.0, 0.0), -math.pi / 2)
    parts = [S.flatten_mesh(S.plane_geometry(3, 3, segments, segments), S.compose_matrix(quaternion=q), i)
             for i in range(copies)]
    # a second, tilted stack crossing the first one: ties between leaves of different subtrees
    q2 = S.quaternion_from_axis_angle((0.0, 0.0, 1.0), 0.6)
    parts += [S.flatten_mesh(S.plane_geometry(2, 2, segments, segments), S.compose_matrix(position=(0, 0.3, 0), quaternion=q2), i)
              for i in range(copies)]
    mats = [dict(color=(0.2 + 0.05 * i, 0.9 - 0.06 * i, 0.5), roughness=1.0, metalness=0.0, specularColor=(1, 1, 1),
                 emissive=(0.03 * (i + 1), 0.5 / (i + 1), 0.1 * (i % 3)), emissiveIntensity=1.0) for i in range(copies)]
    sc = S.Scene(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
                 np.concatenate([p[2] for p in parts]), mats, "coincident sheets")
    sc.build_bvh()
    return sc


def test_equal_t_ties_keep_the_first_visited_leaf_in_every_kernel(gpu_ctx, orc, env):
    """raytrace.wgsl:180 (strict '<'): of equal-t hits the first visited leaf wins.  The
    deferred-leaf kernel (7) tests leaves out of order and resolves ties by visiting rank."""
    sc = _coincident_sheets_scene()
    ctx = gpu_ctx
    pc.upload_scene(ctx, sc, env)
    ctx.set_tile(0, 1, 8)
    w, h = 96, 64
    ctx.resize(w, h)
    frames = (2, 3)
    want = np.zeros((h, w, 4), np.float32)
    osc = pc.oracle_scene(orc, sc, env)
    ocnt = None
    for f in frames:
        img, c = orc.raytrace(osc, pc.rt_uniforms(sc, w, h, frame=f, bounces=3).tobytes(), w, h)
        want = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, img, want)
        ocnt = c if ocnt is None else {k: ocnt[k] + c[k] for k in c}
    picked = set()
    for variant in pc.variants_available(ctx, (2, 4, 7, 8, 9, 10)):
        ctx.set_kernel_variant(variant)
        ctx.reset()
        ctx.reset_counters()
        for f in frames:
            pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=3), pc.acc_uniforms(w, h, f),
                         capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
        got = ctx.read_texture(capi.TEX_ACCUMULATION)
        cnt = ctx.counters()
        assert pc.same_bits(got, want), f"variant {variant}: " + pc.describe_diff(got, want)
        pc.check_counters(cnt, ocnt, culled=variant >= 9, what=f"variant {variant}")
        picked.add(got.tobytes())
    assert len(picked) == 1
    ctx.set_kernel_variant(0)


# ---------------------------------------------------------------- whole passes

FRAME_CASES = [
    # w, h, bounces, spf, aperture, focal, frame, rotation
    (64, 64, 1, 1, 0.0, 1.0, 2, 0.0),
    (64, 64, 4, 1, 0.0, 1.0, 2, 0.0),
    (64, 64, 8, 1, 0.0, 1.0, 3, 0.0),
    (64, 64, 4, 1, 0.05, 4.1, 2, 0.0),
    (64, 64, 4, 3, 0.02, 4.1, 7, 0.7),
    (100, 52, 4, 1, 0.0, 1.0, 2, -2.0),        # ragged: not a multiple of the 8x8 tile
    (256, 256, 4, 1, 0.0, 1.0, 2, 0.0),
    (8, 8, 0, 1, 0.0, 1.0, 2, 0.0),            # maxBounces 0: black
    (1, 1, 4, 1, 0.0, 1.0, 2, 0.0),
]


@pytest.mark.parametrize("variant", [0, 13, 14, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])      # 0 = `auto`, 13 = what it ships, 14 = the 8-wide walk
@pytest.mark.parametrize("case", FRAME_CASES, ids=[f"{c[0]}x{c[1]}-b{c[2]}-s{c[3]}-a{c[4]}" for c in FRAME_CASES])
def test_raytrace_pass_bit_identical(gpu_ctx, orc, demo, env, case, variant):
    w, h, bounces, spf, aperture, focal, frame, rotation = case
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    pc.set_variant_or_skip(ctx, variant)
    ctx.set_storage(capi.STORAGE_F32)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    ctx.reset_counters()
    u = pc.rt_uniforms(demo, w, h, frame=frame, bounces=bounces, spf=spf, aperture=aperture, focal=focal,
                       rotation=rotation)
    pc.gpu_frame(ctx, u)
    got = ctx.read_texture(capi.TEX_OUTPUT)
    cnt = ctx.counters()
    want, ocnt = orc.raytrace(pc.oracle_scene(orc, demo, env), u.tobytes(), w, h)
    assert pc.max_rel_err(got, want) <= REL_TOL
    assert pc.same_bits(got, want), pc.describe_diff(got, want)
    pc.check_counters(cnt, ocnt, culled=variant >= 9 or variant == 0)
    ctx.set_kernel_variant(0)


@pytest.mark.parametrize("variant", [0, 2, 9, 10, 13, 14])
def test_pinhole_camera_with_a_negative_zero_coordinate(gpu_ctx, orc, demo, env, variant):
    """aperture == 0 lets the shipped kernels drop the lens sample's arithmetic (its two rand()
    calls stay): cam_pos + (+-0) is cam_pos -- unless a coordinate of cam_pos is -0, where
    -0 + +0 = +0.  Such a camera must take the general path: image equal to the oracle's, which
    evaluates raytrace.wgsl:447-452 as written."""
    w, h = 96, 64
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_kernel_variant(variant)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    pos = list(demo.camera["position"])
    for k in range(3):
        p = list(pos)
        p[k] = -0.0
        u = pc.rt_uniforms(demo, w, h, frame=5, bounces=4, aperture=0.0, position=p)
        pc.gpu_frame(ctx, u)
        got = ctx.read_texture(capi.TEX_OUTPUT)
        want, _ = orc.raytrace(pc.oracle_scene(orc, demo, env), u.tobytes(), w, h)
        assert pc.same_bits(got, want), pc.describe_diff(got, want)
    ctx.set_kernel_variant(0)
    ctx.resize(64, 64)


@pytest.mark.parametrize("storage", [capi.STORAGE_F32, capi.STORAGE_F16])
@pytest.mark.parametrize("fused", [True, False])
def test_accumulation_over_frames(gpu_ctx, orc, demo, env, storage, fused):
    """3 frames of renderer.render(): frame = 2, 3, 4 (renderer.ts:369-377); the buffer
    then holds sum(c_i)/(n+1)-style running means, with fp16 rounding per step when the
    reference's rgba16float storage is emulated."""
    w = h = 48
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_storage(storage)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    osc = pc.oracle_scene(orc, demo, env)
    acc = np.zeros((h, w, 4), np.float32)
    f16 = storage == capi.STORAGE_F16
    for frame in (2, 3, 4):
        u = pc.rt_uniforms(demo, w, h, frame=frame, bounces=4)
        a = pc.acc_uniforms(w, h, frame)
        if fused:
            pc.gpu_frame(ctx, u, a, capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
        else:
            pc.gpu_frame(ctx, u, a, capi.SUBMIT_RAYTRACE)
            ctx.submit(capi.SUBMIT_ACCUMULATE)
        img, _ = orc.raytrace(osc, u.tobytes(), w, h, store_f16=f16)
        acc = orc.accumulate(a.tobytes(), w, h, img, acc, store_f16=f16)
    got = ctx.read_texture(capi.TEX_ACCUMULATION)
    assert pc.same_bits(got, acc), pc.describe_diff(got, acc)
    assert pc.same_bits(ctx.read_texture(capi.TEX_OUTPUT), acc)     # accumulate.ts:171-175 copy-back
    ctx.set_storage(capi.STORAGE_F32)


def test_deferred_batching_matches_frame_by_frame(gpu_ctx, orc, demo, env):
    """RAYTRACE|ACCUMULATE submits are queued and run as batches of up to 8 consecutive frames
    (one kernel over (frame, tile) jobs + one ordered multi-frame accumulate).  11 frames with
    a camera change in the middle (which splits a batch) must equal the same frames submitted
    with pipelining off (one fused kernel per frame), bit for bit, and the oracle for the
    first frames."""
    w, h = 80, 48
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.set_storage(capi.STORAGE_F32)

    def run(pipelined):
        ctx.set_pipelining(pipelined)
        ctx.resize(w, h)
        ctx.reset_counters()
        for i, frame in enumerate(range(2, 13)):
            u = pc.rt_uniforms(demo, w, h, frame=frame, bounces=4, aperture=0.0 if i < 6 else 0.03, focal=4.1)
            pc.gpu_frame(ctx, u, pc.acc_uniforms(w, h, frame), capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
        return ctx.read_texture(capi.TEX_ACCUMULATION), ctx.counters()

    batched, cb = run(True)
    single, cs = run(False)
    ctx.set_pipelining(True)
    assert pc.same_bits(batched, single), pc.describe_diff(batched, single)
    for k in pc.PATH_COUNTERS:      # (the default walk culls by distance: its box / triangle counts depend on scheduling)
        assert cb[k] == cs[k]
    assert cb["pixels"] == 11 * w * h
    osc = pc.oracle_scene(orc, demo, env)
    acc = np.zeros((h, w, 4), np.float32)
    for i, frame in enumerate(range(2, 13)):
        u = pc.rt_uniforms(demo, w, h, frame=frame, bounces=4, aperture=0.0 if i < 6 else 0.03, focal=4.1)
        img, _ = orc.raytrace(osc, u.tobytes(), w, h)
        acc = orc.accumulate(pc.acc_uniforms(w, h, frame).tobytes(), w, h, img, acc)
    assert pc.same_bits(batched, acc), pc.describe_diff(batched, acc)


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations_shipped_path_equals_per_pixel_kernel(gpu_ctx, demo, env, seed):
    """Random image sizes (1 x 1 upwards, not multiples of the 8 x 8 job tiles), bounce counts from 0, samples per frame,
    thin-lens and pinhole cameras, environment intensity / rotation, frame counts that cut batches anywhere, tile splits,
    both storage formats: the shipped path (auto variant, batched persistent kernel) against the per-pixel kernel that runs
    the WGSL control flow verbatim (variant 1, one fused launch per frame, pipelining off) -- same accumulation bits, same
    path counters.  (The per-pixel kernel itself is held to the oracle by the tests above.)"""
    rng = np.random.default_rng(1000 + seed)
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    w = int(rng.choice([1, 2, 7, 8, 9, 33, 64, 65, 100, 191]))
    h = int(rng.choice([1, 3, 8, 15, 17, 40, 72, 113]))
    nranks = int(rng.choice([1, 1, 2, 3]))
    rank = int(rng.integers(0, nranks))
    block_rows = int(rng.choice([8, 8, 5, 16]))
    storage = capi.STORAGE_F16 if rng.random() < 0.3 else capi.STORAGE_F32
    frames = int(rng.choice([1, 2, 5, 9, 17]))
    spf = int(rng.choice([1, 1, 1, 2, 3]))
    bounces = int(rng.choice([0, 1, 2, 4, 8]))
    kw = dict(bounces=bounces, spf=spf, aperture=float(rng.choice([0.0, 0.0, 0.05])), focal=float(rng.choice([2.0, 4.1])),
              intensity=float(rng.choice([1.0, 0.25])), rotation=float(rng.choice([0.0, 1.3])))
    first = int(rng.integers(1, 1000))
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
    ctx.set_storage(storage)
    ctx.set_tile(rank, nranks, block_rows)
    results = []
    try:
        for variant, pipelined in ((0, True), (1, False)):
            ctx.set_kernel_variant(variant)
            ctx.set_pipelining(pipelined)
            ctx.resize(w, h)
            ctx.reset_counters()
            for f in range(first, first + frames):
                pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=f, **kw), pc.acc_uniforms(w, h, f), mask)
            results.append((ctx.read_texture(capi.TEX_ACCUMULATION), ctx.counters()))
    finally:
        ctx.set_kernel_variant(0)
        ctx.set_pipelining(True)
        ctx.set_storage(capi.STORAGE_F32)
        ctx.set_tile(0, 1, 8)
        ctx.resize(64, 64)
    (a, ca), (b, cb) = results
    what = f"{w}x{h} rank {rank}/{nranks} rows {block_rows} storage {storage} frames {frames} from {first} {kw}"
    assert pc.same_bits(a, b), what + ": " + pc.describe_diff(a, b)
    for k in pc.PATH_COUNTERS:
        assert ca[k] == cb[k], (what, k)


@pytest.mark.parametrize("variant", [0, 4, 7, 9, 10, 11, 12, 13, 14])
def test_every_reference_setting_runs_the_lean_kernel(gpu_ctx, orc, demo, env, variant):
    """samplesPerFrame 1 .. 16 (the reference's slider, main.ts:188), maxBounces 0 .. 10 (main.ts:195), both storage formats,
    queued frames and a launch per frame (pipelining off), the raytrace pass alone: every one of them runs the lean build of
    the state-machine kernel (round 3 sent everything but samplesPerFrame == 1 to a 128-register twin with scratch) and
    renders the oracle's bits.  The sum over a frame's samples lives in the pixel's texel of the frame's radiance slot."""
    w, h = 48, 40
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    osc = pc.oracle_scene(orc, demo, env)
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
    try:
        pc.set_variant_or_skip(ctx, variant)
        want_variant = ctx.active_variant()
        assert want_variant >= 4
        if variant:
            assert want_variant == variant
        for spf, bounces, storage, pipelined, frames, aperture in ((1, 8, capi.STORAGE_F32, True, 3, 0.0), (2, 3, capi.STORAGE_F32, True, 3, 0.0),
                                                                   (5, 2, capi.STORAGE_F16, True, 2, 0.05), (16, 1, capi.STORAGE_F32, True, 1, 0.0),
                                                                   (3, 0, capi.STORAGE_F32, True, 2, 0.0), (1, 0, capi.STORAGE_F16, False, 2, 0.0),
                                                                   (1, 4, capi.STORAGE_F32, False, 3, 0.0), (4, 2, capi.STORAGE_F16, False, 2, 0.0),
                                                                   (0, 2, capi.STORAGE_F32, True, 2, 0.0), (-1, 2, capi.STORAGE_F32, False, 1, 0.0)):
            ctx.set_storage(storage)
            ctx.set_pipelining(pipelined)
            ctx.reset()
            want = np.zeros((h, w, 4), np.float32)
            for f in range(2, 2 + frames):
                u = pc.rt_uniforms(demo, w, h, frame=f, bounces=bounces, spf=spf, aperture=aperture, focal=3.0)
                a = pc.acc_uniforms(w, h, f)
                pc.gpu_frame(ctx, u, a, mask)
                f16 = storage == capi.STORAGE_F16
                img, _ = orc.raytrace(osc, u.tobytes(), w, h, store_f16=f16)
                want = orc.accumulate(a.tobytes(), w, h, img, want, store_f16=f16)
            got = ctx.read_texture(capi.TEX_ACCUMULATION)
            what = f"variant {variant} spf {spf} bounces {bounces} storage {storage} pipelined {pipelined}"
            assert pc.same_bits(got, want), what + ": " + pc.describe_diff(got, want)
            last = ctx.last_launch()
            assert (last["kind"], last["variant"], last["lean"]) == (1, want_variant, True), (what, last)
            # ... and names its instantiation (MI3PT_OPT_LAST_BUILD): this small image's launches run the five-wave build, walk threshold 32
            assert last["waves_per_simd"] == 5 and last["walk_min"] == 32, (what, last)
        # the raytrace pass alone (no accumulate): the frame's radiance, same kernel
        u = pc.rt_uniforms(demo, w, h, frame=9, bounces=4, spf=3)
        pc.gpu_frame(ctx, u)
        img, _ = orc.raytrace(osc, u.tobytes(), w, h)
        assert pc.same_bits(ctx.read_texture(capi.TEX_OUTPUT), img)
        assert ctx.last_launch()["lean"] and ctx.last_launch()["kind"] == 1
    finally:
        ctx.set_kernel_variant(0)
        ctx.set_pipelining(True)
        ctx.set_storage(capi.STORAGE_F32)


def test_accumulate_disabled_passes_frame_through(gpu_ctx, orc, demo, env):
    w = h = 32
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    osc = pc.oracle_scene(orc, demo, env)
    u = pc.rt_uniforms(demo, w, h, frame=5)
    a = pc.acc_uniforms(w, h, 5, enabled=0)
    pc.gpu_frame(ctx, u, a, capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
    img, _ = orc.raytrace(osc, u.tobytes(), w, h)
    want = orc.accumulate(a.tobytes(), w, h, img, np.zeros_like(img))
    assert pc.same_bits(ctx.read_texture(capi.TEX_ACCUMULATION), want)


def test_scaled_subrectangle(gpu_ctx, orc, demo, env):
    """scalingFactor < 1: kernels run on u32(resolution) of a full-size texture with a
    fractional float resolution (raytrace.ts:373, renderer.ts:310-320)."""
    w, h = 90, 50
    res = (w * 0.25, h * 0.25)          # 22.5 x 12.5 -> 22 x 12 pixels
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    u = pc.rt_uniforms(demo, w, h, res=res, aspect=w / h)
    a = layout.UniformBlock(layout.ACCUMULATE_UNIFORMS)
    a.set({"resolution": list(res), "frame": 2, "enabled": 1})
    assert tuple(a.get("resolution")) == (22, 12)
    pc.gpu_frame(ctx, u, a, capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
    osc = pc.oracle_scene(orc, demo, env)
    img, _ = orc.raytrace(osc, u.tobytes(), w, h)
    want = orc.accumulate(a.tobytes(), w, h, img, np.zeros_like(img))
    got = ctx.read_texture(capi.TEX_ACCUMULATION)
    assert pc.same_bits(got, want), pc.describe_diff(got, want)
    assert not got[12:].any() and not got[:, 22:].any()


@pytest.mark.parametrize("denoise,tonemapping,scaling", [(1, 1, 1.0), (0, 1, 1.0), (1, 2, 1.0), (1, 0, 1.0),
                                                         (1, 1, 0.5)])
def test_fullscreen_pass(gpu_ctx, orc, demo, env, denoise, tonemapping, scaling):
    w, h = 72, 40
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    osc = pc.oracle_scene(orc, demo, env)
    acc = np.zeros((h, w, 4), np.float32)
    res = (w * scaling, h * scaling)
    for frame in (2, 3):
        u = pc.rt_uniforms(demo, w, h, frame=frame, res=res, aspect=w / h)
        a = layout.UniformBlock(layout.ACCUMULATE_UNIFORMS)
        a.set({"resolution": list(res), "frame": frame, "enabled": 1})
        pc.gpu_frame(ctx, u, a, capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
        img, _ = orc.raytrace(osc, u.tobytes(), w, h)
        acc = orc.accumulate(a.tobytes(), w, h, img, acc)
    f = pc.fs_uniforms(w, h, scaling, denoise, tonemapping)
    ctx.set_uniforms(capi.PASS_FULLSCREEN, f.tobytes())
    ctx.submit(capi.SUBMIT_FULLSCREEN)
    got_f = ctx.read_texture(capi.TEX_CANVAS)
    got_8 = ctx.read_canvas_rgba8()
    want_f, want_8 = orc.fullscreen(f.tobytes(), acc)
    assert pc.same_bits(got_f, want_f), pc.describe_diff(got_f, want_f)
    assert np.array_equal(got_8, want_8)


@pytest.mark.parametrize("w,h", [(100, 70), (33, 17), (16, 16), (5, 3), (1, 1)])
@pytest.mark.parametrize("scaling,res_scale", [(1.0, 1.0), (0.37, 1.0), (1.0, 0.37), (1.5, 1.0), (1.0, 3.0)])
def test_denoise_over_written_textures(gpu_ctx, orc, w, h, scaling, res_scale):
    """The de-noise pass reads a block's texels from an LDS copy when every tap of a wave lies inside it and from the
    texture otherwise (pt_kernels.hip, "the de-noise pass over an LDS copy"): images smaller than the copy (repeat
    addressing inside it), a scaling factor and resolution uniforms that push taps outside, and texels of every
    magnitude -- noise, flat areas (equal colours: the range weight's exp(0)), zeros, 1e30 and 1e-30 (the weight's
    argument beyond exp's underflow, and products that vanish) -- each against the oracle bit for bit
    (fullscreen.wgsl:53-86, 109-132)."""
    ctx = gpu_ctx
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    rng = np.random.default_rng(w * 1000 + h)
    tex = rng.random((h, w, 4), dtype=np.float32) * np.float32(2.0)
    tex[..., 3] = 1.0
    tex[: h // 3] = np.float32(0.25) + rng.random((h // 3, w, 4), dtype=np.float32) * np.float32(0.05)      # smooth: weights far from 0
    tex[h // 2:, : w // 4, :3] = 0.0
    if h > 4 and w > 4:
        tex[h - 2, w - 2, :3] = np.float32(1e30)
        tex[1, 1, :3] = np.float32(1e-30)
        tex[2, 3, 0] = np.float32(-3.0)
    ctx.write_texture(capi.TEX_ACCUMULATION, tex)
    f = layout.UniformBlock(layout.FULLSCREEN_UNIFORMS)
    f.set({"resolution": [w * res_scale, h * res_scale], "aspect": w / h, "scalingFactor": scaling, "denoise": 1, "tonemapping": 1})
    ctx.set_uniforms(capi.PASS_FULLSCREEN, f.tobytes())
    ctx.submit(capi.SUBMIT_FULLSCREEN)
    got_f = ctx.read_texture(capi.TEX_CANVAS)
    got_8 = ctx.read_canvas_rgba8()
    want_f, want_8 = orc.fullscreen(f.tobytes(), tex)
    assert pc.same_bits(got_f, want_f), pc.describe_diff(got_f, want_f)
    assert np.array_equal(got_8, want_8)
    ctx.resize(64, 64)


@pytest.mark.parametrize("nranks,block_rows", [(2, 8), (4, 8), (8, 8), (3, 5)])
def test_tile_split_reassembles_to_whole_image(gpu_ctx, orc, demo, env, nranks, block_rows):
    """Each rank's compact rows, de-interleaved, must equal the single-GPU image
    (the seed depends only on the global pixel index: raytrace.wgsl:435-436)."""
    w, h = 96, 70
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    u = pc.rt_uniforms(demo, w, h, frame=2, bounces=4)
    a = pc.acc_uniforms(w, h, 2)
    want, _ = orc.raytrace(pc.oracle_scene(orc, demo, env), u.tobytes(), w, h)
    want = orc.accumulate(a.tobytes(), w, h, want, np.zeros_like(want))
    whole = np.zeros((h, w, 4), np.float32)
    total_rays = 0
    for rank in range(nranks):
        ctx.set_tile(rank, nranks, block_rows)
        ctx.resize(w, h)
        ctx.reset_counters()
        pc.gpu_frame(ctx, u, a, capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
        part = ctx.read_texture(capi.TEX_ACCUMULATION)
        from mi3pt_host import tiles
        rows = tiles.local_rows_of(h, rank, nranks, block_rows)
        assert part.shape[0] == len(rows)
        whole[rows] = part
        total_rays += ctx.counters()["rays"]
    assert pc.same_bits(whole, want), pc.describe_diff(whole, want)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)


# ---------------------------------------------------------------- full-size properties

def test_full_hd_properties(gpu_ctx, orc, demo, env):
    """BASELINE.md config 2 size (1920x1080, 8 bounces).  The oracle is too slow for a
    full frame in a unit test, so: (1) both kernel variants agree bit for bit,
    (2) fused == unfused, (3) counter identities hold, (4) an oracle-rendered band of
    rows matches, (5) two tile halves reassemble to the whole."""
    w, h = 1920, 1080
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    u = pc.rt_uniforms(demo, w, h, frame=2, bounces=8)
    a = pc.acc_uniforms(w, h, 2)
    images = {}
    have = pc.variants_available(ctx, (1, 2, 3, 4, 5, 6, 7, 8, 9, 10))
    for variant in have:
        ctx.set_kernel_variant(variant)
        ctx.reset()
        ctx.reset_counters()
        pc.gpu_frame(ctx, u, a, capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
        images[variant] = (ctx.read_texture(capi.TEX_ACCUMULATION), ctx.counters())
    assert all(pc.same_bits(images[v][0], images[2][0]) for v in have)
    pc.check_counters(images[10][1], images[2][1], culled=True, what="wide culling walk")
    pc.check_counters(images[9][1], images[2][1], culled=True, what="distance-culling walk")
    assert images[9][1]["box_tests"] < 0.9 * images[2][1]["box_tests"]      # it does skip boxes
    strip = lambda c: {k: v for k, v in c.items() if k != "reserved"}   # (reserved = fallback-slab count)
    assert all(strip(images[v][1]) == strip(images[2][1]) for v in have if v < 9)
    cnt = images[2][1]
    # the prepared-reciprocal slab test is really in use: only a small share of segments falls back
    assert images[4][1]["reserved"] < 0.05 * cnt["rays"]
    assert cnt["pixels"] == w * h
    assert cnt["rays"] == cnt["hits"] + cnt["misses"]
    assert cnt["pixels"] <= cnt["rays"] <= 8 * cnt["pixels"]
    assert cnt["box_tests"] >= cnt["rays"] and (cnt["box_tests"] - cnt["rays"]) % 2 == 0
    # unfused
    ctx.set_kernel_variant(0)
    ctx.reset()
    pc.gpu_frame(ctx, u, a, capi.SUBMIT_RAYTRACE)
    frame_img = ctx.read_texture(capi.TEX_OUTPUT)
    ctx.submit(capi.SUBMIT_ACCUMULATE)
    assert pc.same_bits(ctx.read_texture(capi.TEX_ACCUMULATION), images[2][0])
    # oracle on a band of 16 rows through the geometry (tile split: 1 block of 16 rows)
    band_rank, nranks, block = 30, 1080 // 16 + 1, 16        # rows 480..495
    part, _ = orc.raytrace(pc.oracle_scene(orc, demo, env), u.tobytes(), w, h, band_rank, nranks, block)
    assert part.shape[0] == 16
    assert pc.same_bits(frame_img[480:496], part), pc.describe_diff(frame_img[480:496], part)
    assert pc.max_rel_err(frame_img[480:496], part) <= REL_TOL


def test_repeated_runs_are_bit_identical(gpu_ctx, demo, env):
    """The persistent kernel hands out jobs dynamically (which wave renders which pixel, and with
    which lane neighbours, changes from run to run); the image must not."""
    w, h = 1280, 720
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    depth = ctx.get_option(capi.OPT_BATCH)
    ctx.set_option(capi.OPT_BATCH, 64)
    runs = []
    for _ in range(3):
        ctx.reset()
        ctx.reset_counters()
        for f in range(2, 82):                   # more than one 64-frame batch: both streams, launch gating
            pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f),
                         capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
        runs.append((ctx.read_texture(capi.TEX_ACCUMULATION).tobytes(), ctx.counters()))
    assert runs[0][0] == runs[1][0] == runs[2][0]
    # the paths are the same; the default walk skips boxes behind the closest hit found SO FAR, which depends on when a
    # lane's parked leaves get their turn (a wave-level vote), so its box / triangle counts may move a little
    for k in pc.PATH_COUNTERS:
        assert runs[0][1][k] == runs[1][1][k] == runs[2][1][k]
    ctx.set_kernel_variant(7)                    # the walk that runs exactly the reference's tests: every counter repeats
    exact = []
    for _ in range(2):
        ctx.reset()
        ctx.reset_counters()
        for f in range(2, 82):
            pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f),
                         capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
        exact.append((ctx.read_texture(capi.TEX_ACCUMULATION).tobytes(), ctx.counters()))
    ctx.set_kernel_variant(0)
    assert exact[0] == exact[1] and exact[0][0] == runs[0][0]
    ctx.set_option(capi.OPT_BATCH, depth)
    ctx.resize(64, 64)


def test_depth_first_relabelling_is_bit_identical(gpu_ctx, demo, env):
    """Round 1 tried node packets numbered in visiting order (node, right subtree, left subtree) and
    triangles stored by leaf rank, saw single-pixel differences on the dragon-class scene, and
    reverted the experiment unexplained.  A relabelling cannot change a pixel.  Re-created here as
    mi3pt_debug_set_packet_layout(1) -- every triangle-indexed array (112-B records, 48-B packets,
    leaf ranks) is permuted consistently -- it is bit-identical to the breadth-first layout, image
    and every counter, in each packet-walking kernel, on the dragon-class scene and on the
    12-way-tie scene: the differences came from the experiment's own (reverted) indexing, not from
    the shipped walk or its tie rule ("Depth-first packet layout")."""
    from mi3pt_host import scenes
    ctx = gpu_ctx
    for sc, (w, h), frames, bounces in ((scenes.dragon_class_scene(), (960, 540), (2, 3, 4), 8),
                                        (_coincident_sheets_scene(), (96, 64), (2, 3), 3)):
        if sc.nodes is None:
            sc.build_bvh()

        def render(variant):
            ctx.set_kernel_variant(variant)
            ctx.reset()
            ctx.reset_counters()
            for f in frames:
                pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=bounces), pc.acc_uniforms(w, h, f),
                             capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
            out = ctx.read_texture(capi.TEX_ACCUMULATION), ctx.counters()
            ctx.set_kernel_variant(0)
            return out

        ctx.set_packet_layout(0)
        pc.upload_scene(ctx, sc, env)
        ctx.set_tile(0, 1, 8)
        ctx.resize(w, h)
        want = {v: render(v) for v in (2, 7)}
        assert pc.same_bits(want[2][0], want[7][0])
        ctx.set_packet_layout(1)
        pc.upload_scene(ctx, sc, env)          # the relabelling applies to what is uploaded from here on
        for v in pc.variants_available(ctx, (1, 2, 3, 4, 6, 7, 8)):
            img, cnt = render(v)
            assert pc.same_bits(img, want[2][0]), f"{sc.name} variant {v}: " + pc.describe_diff(img, want[2][0])
            pc.check_counters(cnt, want[2][1], culled=False, what=f"{sc.name}, relabelled, variant {v}")
        img, cnt = render(0)                   # auto: the culling walk is not offered on a relabelled scene -> 7
        assert pc.same_bits(img, want[2][0])
        pc.check_counters(cnt, want[2][1], culled=False)
        ctx.set_packet_layout(0)
        pc.upload_scene(ctx, sc, env)          # back to the shipped arrangement
        img, cnt = render(7)
        assert pc.same_bits(img, want[2][0])
    ctx.resize(64, 64)


def test_no_cpu_fallback_symbols(built):
    """The product library must not contain an oracle / CPU render path."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in out


def test_submit_frames_equals_frame_by_frame_submits(gpu_ctx, demo, env):
    """mi3pt_submit_frames(mask, n) = n submits with only the frame counter moving
    (renderer.ts:369-377): same image, and both uniform blocks are left at frame + n."""
    w, h = 96, 72
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
    ctx.reset()
    for f in range(2, 25):
        pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=f, bounces=4), pc.acc_uniforms(w, h, f), mask)
    want = ctx.read_texture(capi.TEX_ACCUMULATION)
    ctx.reset()
    ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=2, bounces=4).tobytes())
    ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
    ctx.submit_frames(mask, 20)
    ctx.submit_frames(mask, 3)              # continues at frame 22: the blocks were advanced
    got = ctx.read_texture(capi.TEX_ACCUMULATION)
    assert pc.same_bits(got, want), pc.describe_diff(got, want)
    ctx.resize(64, 64)


def test_present_latest_shows_every_frame_once_the_canvas_is_looked_at(gpu_ctx, orc, demo, env):
    """MI3PT_PRESENT_LATEST: sample frames that also present are queued like any other and the
    canvas is drawn once per launched batch; a canvas read-back (or a FULLSCREEN-only submit)
    first launches what is queued and draws it.  What can be observed equals MI3PT_PRESENT_EXACT,
    which redraws from this very frame on every submit -- and the oracle."""
    w, h = 80, 56
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    everything = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE | capi.SUBMIT_FULLSCREEN
    fs = pc.fs_uniforms(w, h, 1.0, 1, 1)
    canvases = {}
    depth = ctx.get_option(capi.OPT_BATCH)
    ctx.set_option(capi.OPT_BATCH, 64)
    for mode in (capi.PRESENT_EXACT, capi.PRESENT_LATEST):
        ctx.set_present_mode(mode)
        ctx.reset()
        ctx.set_uniforms(capi.PASS_FULLSCREEN, fs.tobytes())
        for f in range(2, 82):              # 80 frames: one full batch + 16 queued frames in LATEST mode
            pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=f, bounces=4), pc.acc_uniforms(w, h, f), everything)
        canvases[mode] = (ctx.read_canvas_rgba8(), ctx.read_texture(capi.TEX_CANVAS), ctx.read_texture(capi.TEX_ACCUMULATION))
        # a FULLSCREEN-only submit (render() after sampling stopped) shows the same
        ctx.submit(capi.SUBMIT_FULLSCREEN)
        assert np.array_equal(ctx.read_canvas_rgba8(), canvases[mode][0])
    ctx.set_present_mode(capi.PRESENT_EXACT)
    ctx.set_option(capi.OPT_BATCH, depth)
    for k in range(3):
        assert np.array_equal(canvases[capi.PRESENT_EXACT][k], canvases[capi.PRESENT_LATEST][k])
    osc = pc.oracle_scene(orc, demo, env)
    acc = np.zeros((h, w, 4), np.float32)
    for f in range(2, 82):
        img, _ = orc.raytrace(osc, pc.rt_uniforms(demo, w, h, frame=f, bounces=4).tobytes(), w, h)
        acc = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, img, acc)
    want_f32, want_8 = orc.fullscreen(fs.tobytes(), acc)
    assert pc.same_bits(canvases[capi.PRESENT_LATEST][1], want_f32)
    assert np.array_equal(canvases[capi.PRESENT_LATEST][0], want_8)
    ctx.resize(64, 64)


@pytest.mark.parametrize("depth", [5, 16, 64])
def test_exact_presentation_shares_launches_without_a_trace(gpu_ctx, orc, demo, env, depth):
    """MI3PT_PRESENT_EXACT draws the canvas behind every presenting frame's own accumulate pass; up to
    MI3PT_OPT_PRESENT_DEPTH such frames share a raytrace launch.  Nothing that can be read differs from depth 1 (a
    launch per frame): canvases and means at two read-backs (the first in the middle of a group), with the fullscreen
    uniforms changing from frame to frame and a frame that does not present in between -- and the last canvas is the
    oracle's (renderer.ts:379-390)."""
    w, h, frames = 80, 56, 23
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    ctx.set_present_mode(capi.PRESENT_EXACT)
    everything = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE | capi.SUBMIT_FULLSCREEN

    def fs_of(f):
        return pc.fs_uniforms(w, h, 1.0, f % 2, 1 + f % 2)

    seen = {}
    for d in (1, depth):
        ctx.set_option(capi.OPT_PRESENT_DEPTH, d)
        assert ctx.get_option(capi.OPT_PRESENT_DEPTH) == d
        ctx.reset()
        ctx.enable_timing(True)
        ctx.raytrace_launch_stats(reset=True)
        shots = []
        for f in range(2, 2 + frames):
            ctx.set_uniforms(capi.PASS_FULLSCREEN, fs_of(f).tobytes())
            mask = everything if f != 11 else capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
            pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=f, bounces=4), pc.acc_uniforms(w, h, f), mask)
            if f in (8, 11, 1 + frames):
                shots.append((ctx.read_canvas_rgba8(), ctx.read_texture(capi.TEX_CANVAS), ctx.read_texture(capi.TEX_ACCUMULATION)))
        seen[d] = shots
        _, launches, nframes = ctx.raytrace_launch_stats()
        assert nframes == frames
        # read-backs after frames 8, 11 and 24 cut the queue: 7 + 3 + 13 frames
        assert launches == sum(-(-n // d) for n in (7, 3, 13))
        ctx.enable_timing(False)
    ctx.set_option(capi.OPT_PRESENT_DEPTH, 16)
    for a, b in zip(seen[1], seen[depth]):
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    # frame 11 did not present: the canvas read behind it is frame 10's (drawn with frame 10's uniforms) ...
    osc = pc.oracle_scene(orc, demo, env)
    acc = np.zeros((h, w, 4), np.float32)
    for f in range(2, 2 + frames):
        img, _ = orc.raytrace(osc, pc.rt_uniforms(demo, w, h, frame=f, bounces=4).tobytes(), w, h)
        acc = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, img, acc)
        if f == 10:
            want10, _ = orc.fullscreen(fs_of(10).tobytes(), acc)
    assert pc.same_bits(seen[depth][1][1], want10)
    # ... and the last one is the last frame's
    want_f32, want_8 = orc.fullscreen(fs_of(1 + frames).tobytes(), acc)
    assert pc.same_bits(seen[depth][2][1], want_f32) and np.array_equal(seen[depth][2][0], want_8)
    assert pc.same_bits(seen[depth][2][2], acc)
    ctx.resize(64, 64)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_presentation_differential_over_random_call_sequences(gpu_ctx, demo, env, seed):
    """Two contexts get the same random sequence of calls -- frames that present, frames that do not, FULLSCREEN-only
    submits, fullscreen uniforms and present modes changing, resets, a resize, read-backs at random points -- one
    launching every presenting frame by itself (MI3PT_OPT_PRESENT_DEPTH 1), one sharing launches (16).  Every read-back
    must agree byte for byte: deferring work never shows."""
    rng = np.random.default_rng(seed)
    own = capi.Context(0)
    try:
        ctxs = (gpu_ctx, own)
        for ctx, depth in zip(ctxs, (1, 16)):
            pc.upload_scene(ctx, demo, env)
            ctx.set_tile(0, 1, 8)
            ctx.set_present_mode(capi.PRESENT_EXACT)
            ctx.set_option(capi.OPT_PRESENT_DEPTH, depth)
            ctx.resize(80, 56)
            ctx.set_uniforms(capi.PASS_FULLSCREEN, pc.fs_uniforms(80, 56, 1.0, 1, 1).tobytes())      # (a shared context arrives with another test's)
        size = [80, 56]
        frame = 2
        everything = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE | capi.SUBMIT_FULLSCREEN
        reads = 0
        for _ in range(120):
            op = rng.choice(["present", "present", "present", "present", "sample", "fs_only", "fs_uniforms", "mode", "read_canvas",
                             "read_accum", "reset", "resize"], p=[0.2, 0.2, 0.15, 0.1, 0.1, 0.03, 0.08, 0.03, 0.05, 0.03, 0.02, 0.01])
            w, h = size
            if op in ("present", "sample"):
                mask = everything if op == "present" else capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
                for ctx in ctxs:
                    pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=frame, bounces=3), pc.acc_uniforms(w, h, frame), mask)
                frame += 1
            elif op == "fs_only":
                for ctx in ctxs:
                    ctx.submit(capi.SUBMIT_FULLSCREEN)
            elif op == "fs_uniforms":
                f = pc.fs_uniforms(w, h, 1.0, int(rng.integers(0, 2)), int(rng.integers(0, 3))).tobytes()
                for ctx in ctxs:
                    ctx.set_uniforms(capi.PASS_FULLSCREEN, f)
            elif op == "mode":
                mode = int(rng.integers(0, 2))
                for ctx in ctxs:
                    ctx.set_present_mode(mode)
            elif op == "read_canvas":
                a, b = (ctx.read_canvas_rgba8() for ctx in ctxs)
                assert np.array_equal(a, b), f"canvas differs at frame {frame}"
                reads += 1
            elif op == "read_accum":
                a, b = (ctx.read_texture(capi.TEX_ACCUMULATION) for ctx in ctxs)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
                reads += 1
            elif op == "reset":
                for ctx in ctxs:
                    ctx.reset()
                frame = 2
            else:
                size = [64, 40] if size == [80, 56] else [80, 56]
                for ctx in ctxs:
                    ctx.resize(*size)
                    ctx.set_uniforms(capi.PASS_FULLSCREEN, pc.fs_uniforms(size[0], size[1], 1.0, 1, 1).tobytes())
                frame = 2
        a, b = (ctx.read_canvas_rgba8() for ctx in ctxs)
        assert np.array_equal(a, b)
        a, b = (ctx.read_texture(capi.TEX_CANVAS) for ctx in ctxs)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        own.close()
        gpu_ctx.set_present_mode(capi.PRESENT_EXACT)
        gpu_ctx.set_option(capi.OPT_PRESENT_DEPTH, 16)
        gpu_ctx.resize(64, 64)


def test_batch_capacity_and_launch_statistics(gpu_ctx, demo, env):
    w, h = 640, 360
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    for nranks, want in ((1, 256), (2, 512), (4, 512), (8, 512), (16, 512)):
        ctx.set_tile(0, nranks, 8)
        ctx.resize(w, h)
        assert ctx.batch_capacity() == want
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    ctx.enable_timing(True)
    ctx.raytrace_launch_stats(reset=True)
    ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=2, bounces=8).tobytes())
    ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
    ctx.submit_frames(capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE, 64)
    total_ms, launches, frames = ctx.raytrace_launch_stats()
    span_ms = ctx.raytrace_launch_span()
    ctx.enable_timing(False)
    assert (launches, frames) == (1, 64)
    # the span runs from the first launch's start to the last one's end: at least one launch long; big launches
    # overlap at their tails (span < sum), small ones like these leave gaps between them (span > sum)
    assert span_ms >= total_ms / 4 and span_ms < 100.0
    ctx.resize(64, 64)


@pytest.mark.parametrize("chunk,group", [(1, 0), (3, -1), (7, 100), (4, 3599), (2, 3600)])
def test_job_order_and_ticket_chunks_lose_and_repeat_nothing(gpu_ctx, demo, env, chunk, group, monkeypatch):
    """The persistent kernel draws its (frame slot, tile) jobs MI3PT_OPT_JOB_CHUNK tickets per atomic
    while the queue is long and singly near its end (640x360 x 16 frames: 57 600 jobs on 3 600
    waves, so the chunked and the single regime both occur), and orders them in groups of
    MI3PT_OPT_JOB_GROUP tiles (0: frame-major; -1: the library's choice; 100: groups that are not
    whole tile rows, with a shorter last group; 3599: a last group of one tile; 3600 = every
    tile: frame-major again).  Whatever the setting: every pixel of every frame exactly once,
    image bit-identical to the context with the defaults."""
    w, h, frames = 640, 360, 16
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
    images = []
    own = capi.Context(0)
    own.set_option(capi.OPT_JOB_CHUNK, chunk)
    own.set_option(capi.OPT_JOB_GROUP, group)
    assert (own.get_option(capi.OPT_JOB_CHUNK), own.get_option(capi.OPT_JOB_GROUP)) == (chunk, group)
    try:
        for ctx in (gpu_ctx, own):
            pc.upload_scene(ctx, demo, env)
            ctx.set_tile(0, 1, 8)
            ctx.resize(w, h)
            ctx.reset()
            ctx.reset_counters()
            ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=2, bounces=4).tobytes())
            ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
            ctx.submit_frames(mask, frames)
            images.append(ctx.read_texture(capi.TEX_ACCUMULATION))
            assert ctx.counters()["pixels"] == w * h * frames
    finally:
        own.close()
    assert pc.same_bits(images[0], images[1]), pc.describe_diff(images[0], images[1])
    gpu_ctx.resize(64, 64)


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_tuned_and_diagnostic_twins_render_the_same_bits(gpu_ctx, demo, env, storage):
    """The shipped batched launch runs a TUNED instantiation of the state-machine kernel: step statistics compiled
    out, the step-voting options as constants, the service step's launch-invariant scalars read back from a block that
    a setup kernel wrote (RtService), 96 vector registers / five waves per SIMD -- and its preconditions are written
    twice, in the host's `tuned` predicate and as constants inside the kernel (round-2 advice).  Binding the
    diagnostic buffer sends the very same job to the generic twin, which computes everything in its prologue: image and
    every counter must agree, for each of the wide walks and for both storage formats (rgba16float is the reference's,
    renderer.ts:102; it runs the tuned twin too)."""
    w, h, frames = 640, 360, 24
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.set_storage(capi.STORAGE_F16 if storage == "f16" else capi.STORAGE_F32)
    ctx.resize(w, h)

    def job(variant):
        ctx.set_kernel_variant(variant)
        ctx.reset()
        ctx.reset_counters()
        ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=2, bounces=6).tobytes())
        ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
        ctx.submit_frames(mask, frames)
        return ctx.read_texture(capi.TEX_ACCUMULATION), ctx.counters()

    try:
        ref, cref = job(2)                      # per-pixel kernel, the WGSL control flow
        for variant in pc.variants_available(ctx, (9, 10, 11, 12, 13, 14)):
            tuned, ct = job(variant)
            ctx.enable_wave_times(True)
            try:
                diag, cd = job(variant)
                stamps = ctx.wave_times()
            finally:
                ctx.enable_wave_times(False)
            assert stamps[:, 2].max() > 0       # the diagnostic twin did run (it stamps its end)
            assert pc.same_bits(tuned, diag), f"variant {variant}: " + pc.describe_diff(tuned, diag)
            assert pc.same_bits(tuned, ref), f"variant {variant}: " + pc.describe_diff(tuned, ref)
            # (box / triangle test counts of the culling walks are not a function of the ray alone: what is skipped depends on
            # the closest hit found SO FAR, i.e. on when the wave ran its triangle steps, which depends on the other lanes)
            for k in pc.PATH_COUNTERS:
                assert ct[k] == cd[k], (variant, k, ct[k], cd[k])
            assert ct["tri_tests"] <= cref["tri_tests"] and cd["tri_tests"] <= cref["tri_tests"]
            for k in pc.PATH_COUNTERS:
                assert ct[k] == cref[k], (variant, k, ct[k], cref[k])
    finally:
        ctx.set_kernel_variant(0)
        ctx.set_storage(capi.STORAGE_F32)
        ctx.resize(64, 64)


def test_triangles_only_upload_after_the_debug_layout_restores_the_uploaded_numbering(gpu_ctx, demo, env):
    """mi3pt_debug_set_packet_layout(1), render, set_packet_layout(0), then upload the TRIANGLES alone: the device still
    held node packets / leaf ranks / the root reference in visiting-order numbering while the triangles were back in
    uploaded order, so leaves referenced the wrong triangles (round-2 advice).  The upload now rebuilds the tree's side."""
    ctx = gpu_ctx
    w, h = 160, 96
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE

    def render(variant):
        ctx.set_kernel_variant(variant)
        ctx.reset()
        for f in (2, 3):
            pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=f, bounces=4), pc.acc_uniforms(w, h, f), mask)
        img = ctx.read_texture(capi.TEX_ACCUMULATION)
        ctx.set_kernel_variant(0)
        return img

    try:
        ctx.set_packet_layout(0)
        pc.upload_scene(ctx, demo, env)
        ctx.set_tile(0, 1, 8)
        ctx.resize(w, h)
        want = render(2)
        ctx.set_packet_layout(1)
        pc.upload_scene(ctx, demo, env)
        assert pc.same_bits(render(7), want)
        ctx.set_packet_layout(0)
        ctx.upload_triangles(demo.triangles)          # triangles only
        for v in (2, 7, 0):
            got = render(v)
            assert pc.same_bits(got, want), f"variant {v}: " + pc.describe_diff(got, want)
        assert ctx.active_variant() >= 9              # the shipped numbering is back: the culling walk is offered again
    finally:
        ctx.set_packet_layout(0)
        pc.upload_scene(ctx, demo, env)
        ctx.resize(64, 64)


@pytest.mark.parametrize("tile,size,mode", [((0, 1), (640, 360), 1), ((1, 3), (328, 200), 1), ((0, 1), (640, 360), 2), ((2, 3), (328, 200), 2)])
def test_cost_ordered_jobs_lose_and_repeat_nothing(gpu_ctx, demo, env, tile, size, mode):
    """MI3PT_OPT_COST_ORDER (an option; off by default -- it did not pay): the first launch for a set of uniforms adds up the
    path segments per 8x8 tile, later launches run the cheapest quarter of the tiles last (mode 1; mode 2: every tile, costliest
    first -- what a launch of a single frame wants) -- a permutation of the tiles,
    chunked into bands, so that a launch's last tickets are its cheapest tiles.  Whatever the order: every pixel of every frame
    exactly once, image bit-identical to a context with the feature off -- through the measuring launch, the ordered
    ones, a camera change (measured again, the other permutation buffer) and back."""
    w, h = size
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
    plain = capi.Context(0)
    gpu_ctx.set_option(capi.OPT_COST_ORDER, mode)
    assert gpu_ctx.get_option(capi.OPT_COST_ORDER) == mode and plain.get_option(capi.OPT_COST_ORDER) == 0
    try:
        for ctx in (gpu_ctx, plain):
            pc.upload_scene(ctx, demo, env)
            ctx.set_tile(tile[0], tile[1], 8)
            ctx.resize(w, h)
            ctx.reset()
            ctx.reset_counters()
        frame, total = 2, 0
        for cam in ((0.0, 1.0, 4.0), (0.0, 1.0, 4.0), (0.0, 1.0, 4.0), (1.5, 0.8, 3.0), (1.5, 0.8, 3.0), (0.0, 1.0, 4.0), (0.0, 1.0, 4.0)):
            n = 24
            imgs = []
            for ctx in (gpu_ctx, plain):
                kw = dict(position=cam, direction=tuple(-np.array(cam) / np.linalg.norm(cam)))
                ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=frame, bounces=6, **kw).tobytes())
                ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, frame).tobytes())
                ctx.submit_frames(mask, n)
                imgs.append(ctx.read_texture(capi.TEX_ACCUMULATION))
            frame += n
            total += n
            assert pc.same_bits(imgs[0], imgs[1]), f"camera {cam}, frame {frame}: " + pc.describe_diff(imgs[0], imgs[1])
            a, b = gpu_ctx.counters(), plain.counters()
            assert a["pixels"] == b["pixels"] == w * gpu_ctx.local_rows * total
            for k in pc.PATH_COUNTERS:
                assert a[k] == b[k], (k, a[k], b[k])
    finally:
        plain.close()
        gpu_ctx.set_option(capi.OPT_COST_ORDER, 0)
        gpu_ctx.set_tile(0, 1, 8)
        gpu_ctx.resize(64, 64)


def test_camera_rays_from_the_precomputed_base_equal_the_in_kernel_ones(gpu_ctx, orc, demo, env):
    """MI3PT_OPT_CAMERA_BASE (round 5): batched launches load the pixel-only part of a camera ray -- uv, cameraToRay's
    direction, cam_pos + dir0 * focalDistance (raytrace.wgsl:219-238, 446) -- from an image formed once per camera instead of
    forming it in every frame.  Same bits with it on and off, against the oracle, and the image follows the camera, the lens,
    the resolution and the tile split (its cache key) from one launch to the next."""
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
    osc = pc.oracle_scene(orc, demo, env)

    def job(w, h, n, **kw):
        ctx.reset()
        ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=2, bounces=4, **kw).tobytes())
        ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
        ctx.submit_frames(mask, n)
        return ctx.read_texture(capi.TEX_ACCUMULATION)

    def oracle(w, h, n, rank=0, nranks=1, **kw):
        acc = np.zeros((orc.tile_local_rows(h, rank, nranks, 8), w, 4), np.float32)
        for f in range(2, 2 + n):
            part, _ = orc.raytrace(osc, pc.rt_uniforms(demo, w, h, frame=f, bounces=4, **kw).tobytes(), w, h, rank, nranks, 8)
            acc = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, part, acc, rank, nranks, 8)
        return acc

    assert ctx.get_option(capi.OPT_CAMERA_BASE) == 1
    ctx.set_tile(0, 1, 8)
    ctx.resize(100, 52)
    cases = [dict(), dict(position=[0.4, 1.3, 3.0]), dict(position=[0.4, 1.3, 3.0], fov=30.0), dict(aperture=0.05, focal=4.1),
             dict(aperture=0.05, focal=2.0), dict(aspect=1.0), dict()]
    for kw in cases:                                    # one context, launch after launch: the key must follow every change
        got = job(100, 52, 5, **kw)
        want = oracle(100, 52, 5, **kw)
        assert pc.same_bits(got, want), (kw, pc.describe_diff(got, want))
    ctx.set_option(capi.OPT_CAMERA_BASE, 0)
    off = job(100, 52, 5)
    ctx.set_option(capi.OPT_CAMERA_BASE, 1)
    assert pc.same_bits(off, oracle(100, 52, 5))
    for spf in (2, 3):                                   # a multi-sample frame's later samples take the same base
        got = job(100, 52, 3, spf=spf)
        ctx.set_option(capi.OPT_CAMERA_BASE, 0)
        off = job(100, 52, 3, spf=spf)
        ctx.set_option(capi.OPT_CAMERA_BASE, 1)
        assert pc.same_bits(got, off), pc.describe_diff(got, off)
    # a rank of a tile split, then another size
    ctx.set_tile(2, 3, 8)
    ctx.resize(100, 52)
    got = job(100, 52, 4)
    assert pc.same_bits(got, oracle(100, 52, 4, 2, 3))
    ctx.set_tile(0, 1, 8)
    ctx.resize(64, 64)
    got = job(64, 64, 4)
    assert pc.same_bits(got, oracle(64, 64, 4))


@pytest.mark.parametrize("variant", [13, 14])
def test_grouping_of_the_tree_into_packets_does_not_change_a_bit(built, orc, demo, env, variant):
    """MI3PT_OPT_COLLAPSE: which descendants of a node its wide packet holds -- the greedy collapse of rounds 2 - 5 (0) or the SAH-optimal one
    (1; round 6) -- for the 4-ary compressed walk (13) and the 8-wide one (14).  Only the leaves' own boxes decide what a ray tests: same image,
    same paths; the option re-runs the scene analysis (own context), also when it changes between two frames of one scene."""
    w, h = 112, 72
    u = pc.rt_uniforms(demo, w, h, frame=4, bounces=6)
    want, ocnt = orc.raytrace(pc.oracle_scene(orc, demo, env), u.tobytes(), w, h)
    with capi.Context(0) as ctx:
        assert ctx.get_option(capi.OPT_COLLAPSE) == -1          # by the packet width: greedy for 4-ary, optimal for 8-ary
        pc.upload_scene(ctx, demo, env)
        ctx.set_kernel_variant(variant)
        ctx.resize(w, h)
        boxes = {}
        for c in (0, 1, -1, 0):
            ctx.set_option(capi.OPT_COLLAPSE, c)
            ctx.reset_counters()
            pc.gpu_frame(ctx, u)
            img, cnt = ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()
            assert ctx.last_launch()["variant"] == variant and ctx.get_option(capi.OPT_COLLAPSE) == c
            assert pc.same_bits(img, want), f"collapse {c}: " + pc.describe_diff(img, want)
            pc.check_counters(cnt, ocnt, culled=True)
            boxes[c] = cnt["box_tests"]
        assert boxes[0] != boxes[1]          # the two groupings ARE different trees of packets (the box-test count shows it)
        with pytest.raises(capi.Mi3ptError):
            ctx.set_option(capi.OPT_COLLAPSE, 2)


@pytest.mark.parametrize("order", [1, 2])
def test_packet_numbering_in_memory_does_not_change_a_bit(built, orc, demo, env, order):
    """MI3PT_OPT_PACKET_ORDER: the 4-ary packets numbered depth-first (1) or in three-level treelets (2) instead of breadth-first.
    The walk follows references: same image, same path counters (its own context: the option re-runs the scene analysis)."""
    w, h = 96, 80
    u = pc.rt_uniforms(demo, w, h, frame=3, bounces=6)
    want, ocnt = orc.raytrace(pc.oracle_scene(orc, demo, env), u.tobytes(), w, h)
    with capi.Context(0) as ctx:
        outs = []
        for o in (0, order):
            ctx.set_option(capi.OPT_PACKET_ORDER, o)
            pc.upload_scene(ctx, demo, env)
            ctx.resize(w, h)
            ctx.reset_counters()
            pc.gpu_frame(ctx, u)
            outs.append((ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()))
            assert ctx.active_variant() == 13 and ctx.get_option(capi.OPT_PACKET_ORDER) == o
        for img, cnt in outs:
            assert pc.same_bits(img, want), pc.describe_diff(img, want)
            pc.check_counters(cnt, ocnt, culled=True)
        for k in pc.PATH_COUNTERS:          # (the box / triangle test counts of a culling walk depend on which rays share a wave: not compared)
            assert outs[0][1][k] == outs[1][1][k]



@pytest.mark.parametrize("six", [0, 1])
def test_five_and_six_wave_builds_render_the_oracles_bits(gpu_ctx, orc, demo, env, six):
    """MI3PT_OPT_SIX_WAVES forced: the small images of this file would only ever run the five-wave build (the six-wave one is taken
    from 1.5 M jobs per launch on) -- here every frame case, multi-sample frames, the lens, F16 storage and a tile split on both."""
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    osc = pc.oracle_scene(orc, demo, env)
    ctx.set_option(capi.OPT_SIX_WAVES, six)
    try:
        ctx.set_tile(0, 1, 8)
        for (w, h, bounces, spf, aperture, focal, frame, rotation) in FRAME_CASES:
            ctx.resize(w, h)
            ctx.reset_counters()
            u = pc.rt_uniforms(demo, w, h, frame=frame, bounces=bounces, spf=spf, aperture=aperture, focal=focal, rotation=rotation)
            pc.gpu_frame(ctx, u)
            got, cnt = ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()
            want, ocnt = orc.raytrace(osc, u.tobytes(), w, h)
            assert pc.same_bits(got, want), ((w, h, bounces, spf), pc.describe_diff(got, want))
            pc.check_counters(cnt, ocnt, culled=True)
            assert ctx.last_launch()["variant"] == 13 and ctx.last_launch()["lean"]
        # batched frames with the running mean, F16 storage, a rank of a split
        w, h = 100, 52
        mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
        for storage, rank, nranks in ((capi.STORAGE_F32, 0, 1), (capi.STORAGE_F16, 0, 1), (capi.STORAGE_F32, 1, 3)):
            ctx.set_storage(storage)
            ctx.set_tile(rank, nranks, 8)
            ctx.resize(w, h)
            ctx.reset()
            ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=2, bounces=6, spf=2, aperture=0.03, focal=3.5).tobytes())
            ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
            ctx.submit_frames(mask, 5)
            got = ctx.read_texture(capi.TEX_ACCUMULATION)
            f16 = storage == capi.STORAGE_F16
            want = np.zeros((orc.tile_local_rows(h, rank, nranks, 8), w, 4), np.float32)
            for f in range(2, 7):
                img, _ = orc.raytrace(osc, pc.rt_uniforms(demo, w, h, frame=f, bounces=6, spf=2, aperture=0.03, focal=3.5).tobytes(), w, h, rank, nranks, 8, store_f16=f16)
                want = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, img, want, rank, nranks, 8, store_f16=f16)
            assert pc.same_bits(got, want), ((storage, rank, nranks), pc.describe_diff(got, want))
    finally:
        ctx.set_option(capi.OPT_SIX_WAVES, -1)
        ctx.set_storage(capi.STORAGE_F32)
        ctx.set_tile(0, 1, 8)
        ctx.resize(64, 64)
