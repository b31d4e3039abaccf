"""The device-built linear BVH (mi3pt_device_build_bvh, csrc/pt_lbvh.hip): an alternative tree in
the reference's 48-byte node records.  Checked for structure (every triangle under exactly one
leaf, children after their parent, parent boxes = union of the children), against the oracle on the
SAME tree (bit-identical, counters included), and against the reference's SAH tree: the reference
walk has no culling, so the image may only differ where two triangles tie in t."""
import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi, layout, scenes

pytestmark = pytest.mark.gpu


def _check_tree(nodes, tris):
    n = len(tris)
    assert len(nodes) == 2 * n - 1
    leaf = nodes["isLeaf"] == 1
    assert leaf.sum() == n and sorted(nodes["triangleIndex"][leaf].tolist()) == list(range(n))
    inner = np.flatnonzero(~leaf)
    idx = np.arange(len(nodes))
    assert (nodes["left"][inner] > idx[inner]).all() and (nodes["right"][inner] > idx[inner]).all()
    assert (nodes["left"][leaf] == -1).all() and (nodes["right"][leaf] == -1).all() and (nodes["triangleIndex"][inner] == -1).all()
    # every node except the root has exactly one parent
    refs = np.concatenate([nodes["left"][inner], nodes["right"][inner]])
    assert sorted(refs.tolist()) == list(range(1, len(nodes)))
    # boxes: leaves bound their triangle exactly, inner nodes are the union of their children
    p = np.stack([tris["aPosition"], tris["bPosition"], tris["cPosition"]], 1)[nodes["triangleIndex"][leaf]]
    assert np.array_equal(nodes["min"][leaf], p.min(1)) and np.array_equal(nodes["max"][leaf], p.max(1))
    l, r = nodes["left"][inner], nodes["right"][inner]
    assert np.array_equal(nodes["min"][inner], np.minimum(nodes["min"][l], nodes["min"][r]))
    assert np.array_equal(nodes["max"][inner], np.maximum(nodes["max"][l], nodes["max"][r]))


@pytest.mark.parametrize("ntris", [1, 2, 3, 17])
def test_tiny_inputs(gpu_ctx, ntris):
    rng = np.random.default_rng(ntris)
    pos = rng.normal(size=(ntris, 3, 3))
    tris = layout.pack_triangles(pos, np.tile([0.0, 0.0, 1.0], (ntris, 3, 1)), np.zeros(ntris, int))
    gpu_ctx.upload_triangles(tris)
    nodes, ms = gpu_ctx.device_build_bvh()
    _check_tree(nodes, tris)


def test_demo_scene_renders_identically(gpu_ctx, orc, demo, env):
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    nodes, ms = ctx.device_build_bvh()
    assert ms > 0
    _check_tree(nodes, demo.triangles)
    w, h = 160, 96
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    u = pc.rt_uniforms(demo, w, h, frame=2, bounces=6)

    def render():
        ctx.reset()
        ctx.reset_counters()
        pc.gpu_frame(ctx, u)
        return ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()

    sah_img, sah_cnt = render()
    ctx.upload_bvh(nodes)
    img, cnt = render()
    want, ocnt = orc.raytrace(orc.OracleScene(demo.triangles, demo.material_bytes, nodes, env), u.tobytes(), w, h)
    assert pc.same_bits(img, want), pc.describe_diff(img, want)
    pc.check_counters(cnt, ocnt, culled=True, what="device-built tree")       # default walk: culls by distance
    # same closest hits as with the reference's tree: same rays, hits and image (no exact ties in this view)
    assert (cnt["rays"], cnt["hits"], cnt["misses"]) == (sah_cnt["rays"], sah_cnt["hits"], sah_cnt["misses"])
    assert pc.same_bits(img, sah_img), pc.describe_diff(img, sah_img)
    assert cnt["stack_overflows"] == 0
    ctx.upload_bvh(demo.nodes)
    ctx.resize(64, 64)


def test_large_mesh_builds_and_renders(gpu_ctx, orc, env):
    sc = scenes.dragon_class_scene()
    ctx = gpu_ctx
    ctx.upload_triangles(sc.triangles)
    ctx.upload_materials(sc.material_bytes)
    ctx.upload_environment(env)
    nodes, ms = ctx.device_build_bvh()
    assert ms < 200.0                               # milliseconds on 870 k triangles (the SAH build: ~0.3 s on 256 host threads)
    _check_tree(nodes, sc.triangles)
    ctx.upload_bvh(nodes)
    w, h = 256, 144
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    u = pc.rt_uniforms(sc, w, h, frame=2, bounces=4)
    ctx.reset_counters()
    pc.gpu_frame(ctx, u)
    img, cnt = ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()
    assert cnt["stack_overflows"] == 0 and cnt["hits"] > 1000
    band, _ = orc.raytrace(orc.OracleScene(sc.triangles, sc.material_bytes, nodes, env), u.tobytes(), w, h, 9, (h + 7) // 8, 8)
    assert pc.same_bits(img[72:80], band), pc.describe_diff(img[72:80], band)
    ctx.resize(64, 64)
