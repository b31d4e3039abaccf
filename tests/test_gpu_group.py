"""Device groups in the library itself (mi3pt_create_group; SURVEY.md 8b "create/destroy a context over N devices", 8e):
a group handle is used through the SAME entry points as a single-device context -- here through the same Python
Context methods -- and must produce the same bits: image, counters, canvas.  The box has one GPU, so the members of
these groups all sit on device 0 (the gather's device-to-device copies are then local; over xGMI they are the same
calls with another source device).  The reference has one adapter and one device (renderer.ts:491-533)."""
import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi

pytestmark = pytest.mark.gpu
MASK = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE


def _job(ctx, sc, env, w, h, frames, per_call):
    pc.upload_scene(ctx, sc, env)
    ctx.resize(w, h)
    ctx.reset_counters()
    ctx.set_uniforms(capi.PASS_FULLSCREEN, pc.fs_uniforms(w, h, 1.0, 1, 1).tobytes())
    f = 2
    while f < 2 + frames:                # Renderer.render(): the frame counter moves, nothing else (renderer.ts:369-377)
        k = min(per_call, 2 + frames - f)
        ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, w, h, frame=f, bounces=5).tobytes())
        ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, f).tobytes())
        if k == 1:
            ctx.submit(MASK | capi.SUBMIT_FULLSCREEN)
        else:
            ctx.submit_frames(MASK, k)
        f += k
    ctx.submit(capi.SUBMIT_FULLSCREEN)
    acc = ctx.read_texture(capi.TEX_ACCUMULATION)
    out = ctx.read_texture(capi.TEX_OUTPUT)
    canvas = ctx.read_canvas_rgba8()
    return acc, out, canvas, ctx.counters()


@pytest.mark.parametrize("members,size,per_call", [(2, (320, 200), 7), (3, (264, 100), 1), (8, (256, 144), 24)])
def test_group_renders_the_single_context_image(gpu_ctx, demo, env, members, size, per_call):
    """2, 3 and 8 members; heights that are not a multiple of 8 x members (a partial last block, members without one,
    members with no rows at all in the last round); per-frame submits that also present, and batched submits."""
    w, h = size
    frames = 24
    gpu_ctx.set_tile(0, 1, 8)
    want = _job(gpu_ctx, demo, env, w, h, frames, per_call)
    with capi.Context(devices=[0] * members) as g:
        assert g.group_size == members
        got = _job(g, demo, env, w, h, frames, per_call)
        for name, a, b in zip(("accumulation", "output", "canvas"), got[:3], want[:3]):
            assert a.shape == b.shape
            assert pc.same_bits(a.astype(np.float32), b.astype(np.float32)), f"{name}: " + pc.describe_diff(a.astype(np.float32), b.astype(np.float32))
        for k in pc.PATH_COUNTERS:
            assert got[3][k] == want[3][k], (k, got[3][k], want[3][k])
        # the members rendered disjoint shares that add up to the image
        pix = [g.member(i).counters()["pixels"] for i in range(members)]
        assert sum(pix) == w * h * frames and max(pix) > 0
        # checkpoint / resume through the group: write the whole image back, render on, same as the single context
        g.write_texture(capi.TEX_ACCUMULATION, want[0])
        gpu_ctx.write_texture(capi.TEX_ACCUMULATION, want[0])
        for ctx in (g, gpu_ctx):
            ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=40, bounces=5).tobytes())
            ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 40).tobytes())
            ctx.submit_frames(MASK, 3)
        a, b = g.read_texture(capi.TEX_ACCUMULATION), gpu_ctx.read_texture(capi.TEX_ACCUMULATION)
        assert pc.same_bits(a, b), pc.describe_diff(a, b)
    gpu_ctx.resize(64, 64)


def test_group_refuses_what_has_no_meaning_for_it(built):
    with capi.Context(devices=[0, 0]) as g:
        for call in (lambda: g.set_tile(0, 2, 8), lambda: g.bind_accumulation(0x1000, 16), lambda: g.set_stream(None)):
            with pytest.raises(capi.Mi3ptError) as e:
                call()
            assert e.value.code == 4                    # MI3PT_ERR_STATE
        with pytest.raises(capi.Mi3ptError):
            g.read_texture(capi.TEX_ACCUMULATION)       # before resize
        g.set_kernel_variant(7)
        assert g.member(0).get_option(capi.OPT_BATCH) == g.member(1).get_option(capi.OPT_BATCH)
    with pytest.raises(capi.Mi3ptError):
        capi.Context(devices=[])


def test_group_compiles_the_scene_once_and_gathers_through_the_host_when_it_must(gpu_ctx, env):
    """Eight members: the host-side scene compile (packets, leaf ranks, the cull analysis with its read-back) runs ONCE per
    scene change, in member 0; the other seven receive device copies (round-3 verdict: it ran eight times).  Then the gather's
    other path: every member's rows through pinned host memory (what a group falls back to without peer access) -- same bits."""
    from mi3pt_host import scenes
    sc = scenes.dragon_class_scene(segments=60)          # ~7 k triangles: a tree the culling walks are offered for
    sc.build_bvh()
    w, h, frames = 200, 116, 10
    gpu_ctx.set_tile(0, 1, 8)
    want = _job(gpu_ctx, sc, env, w, h, frames, 10)
    single = gpu_ctx.get_option(capi.OPT_HOST_ANALYSES)
    with capi.Context(devices=[0] * 8, block_rows=4) as g:
        before = g.get_option(capi.OPT_HOST_ANALYSES)
        got = _job(g, sc, env, w, h, frames, 10)
        assert pc.same_bits(got[0], want[0]), pc.describe_diff(got[0], want[0])
        assert np.array_equal(got[2], want[2])
        # uploads of triangles + BVH, and one cull analysis: what ONE context does for this scene, not eight times that
        compiles = g.get_option(capi.OPT_HOST_ANALYSES) - before
        assert 0 < compiles <= 3, compiles
        assert [g.member(i).get_option(capi.OPT_HOST_ANALYSES) for i in range(1, 8)] == [0] * 7
        assert g.member(3).active_variant() == g.member(0).active_variant() >= 9          # the copies carry the analysis
        # a member's own image has the geometry of its 4-row blocks (round-3 advice: member() assumed 8-row blocks)
        m5 = g.member(5)
        assert m5.local_rows == capi.tile_local_rows(h, 5, 8, 4)
        part = m5.read_texture(capi.TEX_ACCUMULATION)
        from mi3pt_host import tiles
        rows = tiles.local_rows_of(h, 5, 8, 4)              # (the deal goes back and forth: blocks 5, 8 + 2, 16 + 5, 24 + 2 of four rows)
        assert rows[:8] == [20, 21, 22, 23, 40, 41, 42, 43]
        assert part.shape[0] == len(rows) and pc.same_bits(part, want[0][rows])
        # render on; gather staged through the host
        assert g.get_option(capi.OPT_GATHER_STAGED) == 0
        g.set_option(capi.OPT_GATHER_STAGED, 1)
        assert g.get_option(capi.OPT_GATHER_STAGED) == 1
        for ctx in (g, gpu_ctx):
            ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, w, h, frame=30, bounces=5).tobytes())
            ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 30).tobytes())
            ctx.submit_frames(MASK, 4)
        a, b = g.read_texture(capi.TEX_ACCUMULATION), gpu_ctx.read_texture(capi.TEX_ACCUMULATION)
        assert pc.same_bits(a, b), pc.describe_diff(a, b)
        # a scene change: one more compile, again in member 0 only
        g.upload_triangles(sc.triangles)
        g.submit_frames(MASK, 1)
        g.sync()
        assert [g.member(i).get_option(capi.OPT_HOST_ANALYSES) for i in range(1, 8)] == [0] * 7
    assert single > 0
    gpu_ctx.resize(64, 64)
