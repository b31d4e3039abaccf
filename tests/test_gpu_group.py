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
