"""The distance-culling walk (kernel variant 9, the default when the scene allows it) against
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
        for variant in (9, 10):
            got, cgot = _render(ctx, sc, w, h, (2, 3), variant=variant, **kw)
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
        for variant in (9, 10):
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
    for variant in (9, 10):              # (10: a node whose box does not hold its children is not absorbed into a wide packet)
        ctx.set_kernel_variant(variant)
        ctx.reset_counters()
        pc.gpu_frame(ctx, u)
        got, cnt = ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()
        ctx.set_kernel_variant(0)
        assert pc.same_bits(got, want), f"variant {variant}: " + pc.describe_diff(got, want)
        pc.check_counters(cnt, ocnt, culled=True)
    ctx.upload_bvh(demo.nodes)
    ctx.resize(64, 64)


def test_demo_scene_1080p_and_the_share_of_boxes_skipped(gpu_ctx, demo, env):
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    w, h = 1920, 1080
    ctx.resize(w, h)
    ref, cref = _render(ctx, demo, w, h, range(2, 8), variant=7)
    for variant in (9, 10):
        got, cgot = _render(ctx, demo, w, h, range(2, 8), variant=variant)
        assert pc.same_bits(got, ref), pc.describe_diff(got, ref)
        pc.check_counters(cgot, cref, culled=True)
        assert cgot["box_tests"] < cref["box_tests"]
    ctx.resize(64, 64)
