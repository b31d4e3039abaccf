"""The distance-culling walks (kernel variants 9 and 10 .. 12; one of 10 / 11 / 12 is the default when the scene allows
it: 11 and 12 decide boxes from approximate quotients wherever that provably gives the exact test's answer) against
the walks that execute exactly the reference's tests (raytrace.wgsl:118-203 has no bound by the
current hit).  Contract: images bit-identical, the same paths (rays / hits /
misses / pixels), fewer box and triangle tests.  The scenes here are chosen to stress the
margin: grazing views over medium-size triangles, slivers, huge and tiny triangles side by
side, a camera inside the geometry, a tree whose boxes do not bound its triangles."""
import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi, layout, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[(-1, 0), (0, 0), (1, 0), (1, 44), (0, 44)],
                ids=["waves-auto", "five-waves", "six-waves", "six-waves-deep-build", "five-waves-deep-build"])
def waves_per_simd(request, gpu_ctx):
    """Every test of this file with the shipped walk's five- and six-wave builds forced (MI3PT_OPT_SIX_WAVES): the six-wave build has a
    19-entry LDS stack, so the stack tests below cross its LDS part at other depths; and with the walk threshold of very large trees
    (44: their own instantiations -- the six-wave one keeps a 25-entry stack and the parked path state in memory)."""
    six, walk_min = request.param
    gpu_ctx.set_option(capi.OPT_SIX_WAVES, six)
    gpu_ctx.set_option(capi.OPT_WALK_MIN, walk_min)
    yield request.param
    gpu_ctx.set_option(capi.OPT_SIX_WAVES, -1)
    gpu_ctx.set_option(capi.OPT_WALK_MIN, 0)


def _render(ctx, sc, w, h, frames, variant, bounces=8, **kw):
    ctx.set_kernel_variant(variant)
    ctx.reset()
    ctx.reset_counters()
    for f in frames:
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=bounces, **kw), pc.acc_uniforms(w, h, f),
                     capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
    img = ctx.read_texture(capi.TEX_ACCUMULATION)
    cnt = ctx.counters()
    ctx.set_kernel_variant(0)
    return img, cnt


def _soup(seed, n, size_lo, size_hi, spread, sliver=False):
    """Random triangle soup with log-uniform sizes; optional slivers (aspect down to 1e-3)."""
    rng = np.random.default_rng(seed)
    centre = rng.normal(size=(n, 3)) * spread
    size = 10 ** rng.uniform(np.log10(size_lo), np.log10(size_hi), n)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    v = rng.normal(size=(n, 3)); v -= (v * u).sum(1)[:, None] * u; v /= np.linalg.norm(v, axis=1)[:, None]
    asp = 10 ** rng.uniform(-3, 0, n) if sliver else rng.uniform(0.3, 1.0, n)
    pos = np.stack([centre, centre + u * size[:, None], centre + v * (size * asp)[:, None]], 1)
    nrm = np.repeat(np.cross(u, v)[:, None, :], 3, 1)
    mats = [dict(color=(0.8, 0.7, 0.6), roughness=1.0, metalness=0.3, specularColor=(1, 1, 1)),
            dict(color=(0.2, 0.5, 0.9), roughness=0.2, metalness=0.9, specularColor=(1, 1, 1),
                 emissive=(0.4, 0.3, 0.2), emissiveIntensity=0.5)]
    sc = scenes.Scene(pos, nrm, rng.integers(0, 2, n), mats, f"soup-{seed}")
    sc.build_bvh()
    return sc


SOUPS = {
    "small triangles": dict(seed=3, n=40000, size_lo=0.004, size_hi=0.04, spread=0.6),
    "medium triangles near the E cap": dict(seed=4, n=6000, size_lo=0.05, size_hi=0.4, spread=1.5),
    "tiny next to huge": dict(seed=5, n=20000, size_lo=0.0005, size_hi=3.0, spread=1.0),
    "slivers": dict(seed=6, n=20000, size_lo=0.01, size_hi=0.5, spread=0.8, sliver=True),
}


@pytest.mark.parametrize("name", list(SOUPS))
def test_soups_culled_walk_is_bit_identical(gpu_ctx, env, name):
    sc = _soup(**SOUPS[name])
    ctx = gpu_ctx
    pc.upload_scene(ctx, sc, env)
    ctx.set_tile(0, 1, 8)
    w, h = 256, 160
    ctx.resize(w, h)
    for cam in ((0.0, 0.3, 2.5), (0.05, 0.02, 0.1), (3.0, 0.0, 0.0)):         # outside, inside the cloud, from the side
        kw = dict(position=cam, direction=tuple(-np.array(cam) / np.linalg.norm(cam)))
        ref, cref = _render(ctx, sc, w, h, (2, 3), variant=2, **kw)
        for variant in pc.variants_available(ctx, (9, 10, 11, 12, 13, 14)):
            ctx.set_kernel_variant(variant)
            assert ctx.active_variant() == variant
            got, cgot = _render(ctx, sc, w, h, (2, 3), variant=variant, **kw)
            if variant in (13, 14):
                assert ctx.last_launch() == dict(ctx.last_launch(), kind=1, variant=variant, lean=True)      # the compressed-wide / 8-wide kernel did run
            assert pc.same_bits(got, ref), f"{name} camera {cam} variant {variant}: " + pc.describe_diff(got, ref)
            pc.check_counters(cgot, cref, culled=True, what=name)
    ctx.resize(64, 64)


def test_grazing_views_over_a_tessellated_floor(gpu_ctx, env):
    """Rays almost parallel to many medium-size triangles: the case where Moller-Trumbore's t is
    least accurate and the margin has to be widest."""
    q = scenes.quaternion_from_axis_angle((1.0, 0.0, 0.0), -np.pi / 2)
    floor = scenes.flatten_mesh(scenes.plane_geometry(8, 8, 60, 60), scenes.compose_matrix(quaternion=q), 0)
    ball = scenes.flatten_mesh(scenes.sphere_geometry(0.4, 48, 32), scenes.compose_matrix(position=(0.0, 0.4, 0.0)), 1)
    mats = [scenes.WHITE, dict(color=(0.9, 0.9, 0.9), roughness=0.05, metalness=1.0, specularColor=(1, 1, 1))]
    sc = scenes.Scene(np.concatenate([floor[0], ball[0]]), np.concatenate([floor[1], ball[1]]),
                      np.concatenate([floor[2], ball[2]]), mats, "grazing floor")
    sc.build_bvh()
    ctx = gpu_ctx
    pc.upload_scene(ctx, sc, env)
    ctx.set_tile(0, 1, 8)
    w, h = 320, 200
    ctx.resize(w, h)
    for height in (1e-1, 1e-3, 1e-5, 2e-7):
        cam = (0.0, height, 3.9)
        kw = dict(position=cam, direction=(0.0, 0.0, -1.0), fov=60.0)
        ref, cref = _render(ctx, sc, w, h, (2, 3, 4), variant=7, **kw)
        for variant in pc.variants_available(ctx, (9, 10, 11, 12, 13, 14)):
            got, cgot = _render(ctx, sc, w, h, (2, 3, 4), variant=variant, **kw)
            assert pc.same_bits(got, ref), f"camera height {height} variant {variant}: " + pc.describe_diff(got, ref)
            pc.check_counters(cgot, cref, culled=True, what=f"height {height}")
    ctx.resize(64, 64)


def test_boxes_that_do_not_bound_their_triangles_are_never_skipped(gpu_ctx, orc, demo, env):
    """The reference does not care whether a box bounds anything (a hit is a hit); the culled walk
    does.  Shrinking and shifting boxes of the demo tree must switch the distance bound off where
    it no longer holds: the image still equals the oracle's on the same (broken) tree."""
    nodes = demo.nodes.copy()
    raw = nodes.view(np.uint8).reshape(len(nodes), 48).copy()
    f = raw.view(np.float32).reshape(len(nodes), 12)
    rng = np.random.default_rng(5)
    pick = rng.choice(np.arange(1, len(nodes)), 400, replace=False)
    f[pick, 0:3] += rng.uniform(0.0, 0.05, (400, 3)).astype(np.float32)       # min up
    f[pick, 4:7] += rng.uniform(-0.02, 0.05, (400, 3)).astype(np.float32)     # max moved
    broken = raw.view(nodes.dtype).reshape(len(nodes))
    ctx = gpu_ctx
    ctx.upload_bvh(broken)
    ctx.upload_triangles(demo.triangles)
    ctx.upload_materials(demo.material_bytes)
    ctx.upload_environment(env)
    ctx.set_tile(0, 1, 8)
    w, h = 128, 96
    ctx.resize(w, h)
    u = pc.rt_uniforms(demo, w, h, frame=2, bounces=6)
    want, ocnt = orc.raytrace(orc.OracleScene(demo.triangles, demo.material_bytes, broken, env), u.tobytes(), w, h)
    for variant in pc.variants_available(ctx, (9, 10, 11, 12, 13, 14)):              # (10: a node whose box does not hold its children is not absorbed into a wide packet)
        ctx.set_kernel_variant(variant)
        ctx.reset_counters()
        pc.gpu_frame(ctx, u)
        got, cnt = ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()
        ctx.set_kernel_variant(0)
        assert pc.same_bits(got, want), f"variant {variant}: " + pc.describe_diff(got, want)
        pc.check_counters(cnt, ocnt, culled=True)
    ctx.upload_bvh(demo.nodes)
    ctx.resize(64, 64)


def _comb_scene(depth=40):
    """A hand-made tree that drives the near-first walks' node stack past its LDS part: a spine
    of `depth` internal nodes, each with an internal side branch (two triangles) that lies BEHIND
    the rest of the spine as seen from the camera.  All boxes span the whole x / y range, so a ray
    that misses every triangle still enters every box: the walk descends the (nearer) spine first
    and one side branch per level piles up -- `depth` entries, where the LDS holds 24."""
    pos, leaves = [], []
    z_of = lambda k: -40.0 + k               # side branch k sits at z_of(k): farther from the camera (z = +5) for small k
    for k in range(depth + 1):
        for j in range(2):
            x0 = 0.25 + 0.3 * j
            pos.append([[x0, -0.4, z_of(k) + 0.02 * j], [x0 + 0.25, -0.4, z_of(k) + 0.02 * j], [x0 + 0.1, 0.5, z_of(k) + 0.02 * j]])
    pos = np.array(pos, np.float64)
    ntri = len(pos)
    tris = layout.pack_triangles(pos, np.tile(np.array([0.0, 0.0, 1.0]), (ntri, 3, 1)), np.zeros(ntri, int))
    nodes = np.zeros(2 * ntri - 1, layout.BVH_NODE)

    def put(i, zlo, zhi, leaf=-1, left=-1, right=-1):
        nodes[i]["min"], nodes[i]["max"] = (-1.0, -1.0, zlo - 0.1), (1.0, 1.0, zhi + 0.1)
        nodes[i]["isLeaf"], nodes[i]["left"], nodes[i]["right"], nodes[i]["triangleIndex"] = (1 if leaf >= 0 else 0), left, right, leaf

    idx = 0
    for k in range(depth):                   # spine node k at idx: left = side branch k (idx+1, leaves idx+2, idx+3), right = idx+4
        put(idx, z_of(k), z_of(depth), left=idx + 1, right=idx + 4)
        put(idx + 1, z_of(k), z_of(k), left=idx + 2, right=idx + 3)
        put(idx + 2, z_of(k), z_of(k), leaf=2 * k)
        put(idx + 3, z_of(k), z_of(k), leaf=2 * k + 1)
        idx += 4
    put(idx, z_of(depth), z_of(depth), left=idx + 1, right=idx + 2)
    put(idx + 1, z_of(depth), z_of(depth), leaf=2 * depth)
    put(idx + 2, z_of(depth), z_of(depth), leaf=2 * depth + 1)
    assert idx + 3 == len(nodes)
    mats = layout.pack_materials([dict(color=(0.9, 0.8, 0.7), roughness=1.0, metalness=0.5, specularColor=(1, 1, 1))])
    return tris, mats, nodes


def test_node_stack_beyond_its_lds_part(gpu_ctx, orc, env):
    """The near-first walks keep 24 node entries per lane in LDS and up to 32 more in the wave's overflow
    slice (pt_kernels.h SM_CULL_STACK_MAX).  On the comb scene the stack reaches ~40 (binary packets) /
    ~42 (wide packets): images and paths must still equal the oracle's."""
    tris, mats, nodes = _comb_scene()
    ctx = gpu_ctx
    ctx.upload_bvh(nodes)
    ctx.upload_triangles(tris)
    ctx.upload_materials(mats)
    ctx.upload_environment(env)
    ctx.set_tile(0, 1, 8)
    w, h = 96, 64
    ctx.resize(w, h)

    class Cam:
        camera = dict(position=(0.0, 0.0, 5.0), fov=3.0, focalDistance=1.0, aperture=0.0)

        @staticmethod
        def camera_direction():
            return (0.0, 0.0, -1.0)

    u = pc.rt_uniforms(Cam, w, h, frame=2, bounces=4)
    want, ocnt = orc.raytrace(orc.OracleScene(tris, mats, nodes, env), u.tobytes(), w, h)
    assert ocnt["stack_overflows"] == 0 and 0 < ocnt["hits"] < ocnt["rays"]
    assert ocnt["box_tests"] > 100 * ocnt["rays"]             # every ray enters (almost) every box
    for variant in pc.variants_available(ctx, (9, 10, 11, 12, 13, 14)):
        ctx.set_kernel_variant(variant)
        assert ctx.active_variant() == variant            # no silent fall-back: this tree admits both walks
        ctx.reset_counters()
        pc.gpu_frame(ctx, u)
        got, cnt = ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()
        ctx.set_kernel_variant(0)
        assert pc.same_bits(got, want), f"variant {variant}: " + pc.describe_diff(got, want)
        pc.check_counters(cnt, ocnt, culled=True, what=f"variant {variant}")
    ctx.resize(64, 64)


def _deep_eight_wide_scene(depth):
    """A hand-made tree whose 8-ary collapse (kernel variant 14) is `depth` + 2 packet levels deep AND whose walk piles up one node
    entry per level: a spine of `depth` internal nodes; spine node k has a side subtree of eight triangles whose internal boxes are
    as large as the spine node's own (so the collapse -- largest area first -- opens THEM and leaves the next spine node a child
    packet) and lie, by their centres, behind the rest of the spine as seen from the camera (so the walk descends the spine first
    and one internal child of every level stays in its entry).  Every box spans the whole x / y range: every ray enters every box."""
    z_of = lambda k: -40.0 + k
    pos = []
    for k in range(depth + 1):
        for j in range(8 if k < depth else 2):
            x0 = -0.9 + 0.22 * j
            pos.append([[x0, -0.4, z_of(k) + 0.01 * j], [x0 + 0.2, -0.4, z_of(k) + 0.01 * j], [x0 + 0.1, 0.5, z_of(k) + 0.01 * j]])
    pos = np.array(pos, np.float64)
    ntri = len(pos)
    tris = layout.pack_triangles(pos, np.tile(np.array([0.0, 0.0, 1.0]), (ntri, 3, 1)), np.zeros(ntri, int))
    p32 = pos.astype(np.float32)
    nodes = np.zeros(2 * ntri - 1, layout.BVH_NODE)

    def put(i, mn, mx, leaf=-1, left=-1, right=-1):
        nodes[i]["min"], nodes[i]["max"] = mn, mx
        nodes[i]["isLeaf"], nodes[i]["left"], nodes[i]["right"], nodes[i]["triangleIndex"] = (1 if leaf >= 0 else 0), left, right, leaf

    def put_leaf(i, t):
        put(i, tuple(p32[t].min(0)), tuple(p32[t].max(0)), leaf=t)

    zhi = z_of(depth) + 0.2
    idx, t = 0, 0
    for k in range(depth):
        spine_box = ((-1.0, -1.0, z_of(k) - 0.1), (1.0, 1.0, zhi))
        fat = ((-1.0, -1.0, z_of(k) - 0.1), (1.0, 1.0, zhi - 0.01))          # the side subtree's internal boxes: centre a little farther from the camera
        put(idx, *spine_box, left=idx + 1, right=idx + 16)
        # side subtree: root idx+1; halves idx+2, idx+9; quarters idx+3, idx+6, idx+10, idx+13; leaves behind each quarter
        put(idx + 1, *fat, left=idx + 2, right=idx + 9)
        for h, base in enumerate((idx + 2, idx + 9)):
            put(base, *fat, left=base + 1, right=base + 4)
            for q, qb in enumerate((base + 1, base + 4)):
                put(qb, *fat, left=qb + 1, right=qb + 2)
                put_leaf(qb + 1, t); put_leaf(qb + 2, t + 1)
                t += 2
        idx += 16
    put(idx, (-1.0, -1.0, z_of(depth) - 0.1), (1.0, 1.0, zhi), left=idx + 1, right=idx + 2)
    put_leaf(idx + 1, t); put_leaf(idx + 2, t + 1)
    assert idx + 3 == len(nodes) and t + 2 == ntri
    mats = layout.pack_materials([dict(color=(0.9, 0.8, 0.7), roughness=1.0, metalness=0.5, specularColor=(1, 1, 1))])
    return tris, mats, nodes


@pytest.mark.parametrize("depth,want_variant", [(6, 14), (24, 14), (40, 9)])
def test_eight_wide_node_stack_beyond_its_lds_part(gpu_ctx, orc, env, depth, want_variant):
    """The 8-wide walk (variant 14) keeps 8 .. 11 64-bit node entries per lane in LDS and up to 22 more in the wave's overflow slice;
    its stack holds one entry per packet level.  On this tree the walk really stacks `depth` entries: 6 stays in LDS, 24 uses the
    overflow slice in every build (and is a tree the 4-ary walks do not admit: their order-independent bound is three entries per
    level -- the 8-wide packets have their own preconditions), 40 is more levels than LDS and the slice hold together -- the context must
    then hand the scene to another walk (here the binary culling walk, 9).
    Images and paths equal the oracle's either way."""
    tris, mats, nodes = _deep_eight_wide_scene(depth)
    ctx = gpu_ctx
    ctx.upload_bvh(nodes)
    ctx.upload_triangles(tris)
    ctx.upload_materials(mats)
    ctx.upload_environment(env)
    ctx.set_tile(0, 1, 8)
    w, h = 96, 64
    ctx.resize(w, h)

    class Cam:
        camera = dict(position=(0.0, 0.0, 5.0), fov=3.0, focalDistance=1.0, aperture=0.0)

        @staticmethod
        def camera_direction():
            return (0.0, 0.0, -1.0)

    u = pc.rt_uniforms(Cam, w, h, frame=2, bounces=4)
    want, ocnt = orc.raytrace(orc.OracleScene(tris, mats, nodes, env), u.tobytes(), w, h)
    assert ocnt["stack_overflows"] == 0 and 0 < ocnt["hits"] < ocnt["rays"]
    ctx.set_option(capi.OPT_COLLAPSE, 0)          # the greedy grouping follows the areas this tree is built around (the optimal one packs it flat)
    ctx.set_kernel_variant(14)
    try:
        assert ctx.active_variant() == want_variant
        ctx.reset_counters()
        pc.gpu_frame(ctx, u)
        got, cnt = ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()
        assert ctx.last_launch()["variant"] == want_variant
        assert ctx.last_launch()["lean"] or want_variant != 14          # (the fall-back walk has no lean build for the deep-walk threshold)
        assert pc.same_bits(got, want), pc.describe_diff(got, want)
        pc.check_counters(cnt, ocnt, culled=True, what=f"variant {want_variant}, depth {depth}")
        if want_variant == 14:
            # every ray enters every box of the spine: at least one node step (eight box tests) per level
            assert cnt["box_tests"] >= 8 * depth * ocnt["rays"] * 0.9
    finally:
        ctx.set_kernel_variant(0)
        ctx.set_option(capi.OPT_COLLAPSE, -1)
        ctx.resize(64, 64)


def test_demo_scene_1080p_and_the_share_of_boxes_skipped(gpu_ctx, demo, env):
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    w, h = 1920, 1080
    ctx.resize(w, h)
    ref, cref = _render(ctx, demo, w, h, range(2, 8), variant=7)
    for variant in pc.variants_available(ctx, (9, 10, 11, 12, 13, 14)):
        got, cgot = _render(ctx, demo, w, h, range(2, 8), variant=variant)
        assert pc.same_bits(got, ref), pc.describe_diff(got, ref)
        pc.check_counters(cgot, cref, culled=True)
        if variant != 14:
            assert cgot["box_tests"] < cref["box_tests"]       # (this view: both walks test fewer boxes than the reference; the 8-wide walk tests eight per step)
    ctx.resize(64, 64)


def test_nan_rays_take_the_known_answer(gpu_ctx, orc, demo, env):
    """A triangle whose vertex normals are zero gives normalize(0) = NaN as the shading normal, the
    bounce direction is NaN, and such a ray passes EVERY box test of the reference's walk (min / max
    drop the NaNs) while no triangle can accept it: a walk of the whole tree for a miss (18 s per
    ray on a 10 M-triangle scene).  The culling walks return the miss at once; the pixels (NaNs
    included) and the path counters equal the oracle's and the per-pixel kernel's, which walk it."""
    nrm = demo.normals.copy()
    nrm[::3] = 0.0                          # every third triangle: zero normals at its three vertices
    sc = scenes.Scene(demo.positions, nrm, demo.material_index, demo.materials, "demo with zero normals")
    sc.build_bvh(nthreads=2)
    w, h, frames = 160, 96, (2, 3)
    ctx = gpu_ctx
    pc.upload_scene(ctx, sc, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    osc = pc.oracle_scene(orc, sc, env)
    acc = np.zeros((h, w, 4), np.float32)
    orays = 0
    for f in frames:
        img, oc = orc.raytrace(osc, pc.rt_uniforms(sc, w, h, frame=f, bounces=4).tobytes(), w, h)
        acc = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, img, acc)
        orays += oc["rays"]
    assert np.isnan(acc).any()              # the case is there
    ref, cref = _render(ctx, sc, w, h, frames, 2, bounces=4)
    assert pc.same_bits(ref, acc), pc.describe_diff(ref, acc)
    assert cref["rays"] == orays
    for variant in pc.variants_available(ctx, (9, 10, 11, 12, 13, 14)):
        got, cgot = _render(ctx, sc, w, h, frames, variant, bounces=4)
        assert ctx.active_variant() in (variant, 10, 11, 12, 13, 14)
        assert pc.same_bits(got, acc), pc.describe_diff(got, acc)
        for k in pc.PATH_COUNTERS:
            assert cgot[k] == cref[k], (variant, k, cgot[k], cref[k])
        assert cgot["box_tests"] < 0.7 * cref["box_tests"]       # the NaN rays alone walk ~4 000 boxes each in the reference
    ctx.resize(64, 64)
