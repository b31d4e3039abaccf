"""The contiguous split (mi3pt_set_rows): a context renders a band of rows [first, first + n) of the image instead of
round-robin 8-row tiles; bands cut by measured cost (mi3pt_measure_tile_cost + tiles.balanced_bands).  Any partition of
the rows must render the whole image's bits (the seed comes from the global pixel index, raytrace.wgsl:435-436)."""
import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi, tiles

pytestmark = pytest.mark.gpu
MASK = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE


def _render(ctx, sc, w, h, frames, bounces=5):
    ctx.resize(w, h)
    ctx.reset_counters()
    ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, w, h, frame=2, bounces=bounces).tobytes())
    ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
    ctx.submit_frames(MASK, frames)
    return ctx.read_texture(capi.TEX_ACCUMULATION), ctx.read_texture(capi.TEX_OUTPUT), ctx.counters()


@pytest.mark.parametrize("storage", [capi.STORAGE_F32, capi.STORAGE_F16])
def test_any_partition_into_bands_renders_the_whole_image(gpu_ctx, demo, env, storage):
    w, h, frames = 152, 99, 9
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_storage(storage)
    try:
        ctx.set_tile(0, 1, 8)
        ctx.set_rows(0, -1)
        want, want_out, cw = _render(ctx, demo, w, h, frames)
        for bounds in ([0, 40, 99], [0, 5, 5, 41, 99], [0, 1, 98, 99, 99], [0, 13, 26, 39, 52, 65, 78, 91, 99]):
            parts, outs, rays, pixels = [], [], 0, 0
            for r in range(len(bounds) - 1):
                ctx.set_rows(bounds[r], bounds[r + 1] - bounds[r])
                acc, out, c = _render(ctx, demo, w, h, frames)
                assert acc.shape == (bounds[r + 1] - bounds[r], w, 4) == out.shape
                parts.append(acc); outs.append(out)
                rays += c["rays"]; pixels += c["pixels"]
                if acc.shape[0]:
                    assert ctx.last_launch()["lean"]
            whole = tiles.stack_bands(parts, bounds)
            assert pc.same_bits(whole, want), f"bands {bounds}: " + pc.describe_diff(whole, want)
            assert pc.same_bits(tiles.stack_bands(outs, bounds), want_out)
            assert (rays, pixels) == (cw["rays"], cw["pixels"])
        # a band past the image's bottom holds nothing; the fullscreen pass needs the whole image
        ctx.set_rows(120, 30)
        ctx.resize(w, h)
        assert ctx.local_rows == 0 and ctx.read_texture(capi.TEX_ACCUMULATION).shape[0] == 0
        ctx.set_rows(8, 16)
        ctx.resize(w, h)
        with pytest.raises(capi.Mi3ptError):
            ctx.submit(capi.SUBMIT_FULLSCREEN)
        with pytest.raises(capi.Mi3ptError):
            ctx.set_rows(-1, 4)
    finally:
        ctx.set_rows(0, -1)
        ctx.set_storage(capi.STORAGE_F32)
        ctx.resize(64, 64)


def test_measured_cost_cuts_balanced_bands(gpu_ctx, demo, env):
    """mi3pt_measure_tile_cost: one frame of the whole image, what each 8x8 tile's paths cost -- repeatable to a fraction of a
    per cent, not to the unit (how many boxes and triangles a culling walk tests depends on when its closest hit so far shrinks,
    i.e. on which rays share its wave: images are deterministic, test counts are not), so ONE rank measures and the bands are
    broadcast; the bands it cuts reassemble to the whole image."""
    w, h, frames = 320, 184, 6
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.set_rows(0, -1)
    try:
        want, _, cw = _render(ctx, demo, w, h, frames)
        ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(demo, w, h, frame=2, bounces=5).tobytes())
        cost = ctx.measure_tile_cost()
        assert cost.shape == (23, 40) and cost.min() >= 10 * 64 * 0 and cost.sum() > 0
        again = ctx.measure_tile_cost().astype(np.int64)                      # the same frame: the same costs within a per cent per tile row
        assert np.abs(again.sum(axis=1) - cost.sum(axis=1).astype(np.int64)).max() <= 0.01 * cost.sum(axis=1).max()
        assert not ctx.last_launch()["lean"]                                  # (the diagnostic twin does the measuring)
        # the rows that show the models cost more than the top of the image
        assert cost[:3].mean() < 0.8 * cost[8:16].mean()
        # measuring does not touch the accumulation image
        assert pc.same_bits(ctx.read_texture(capi.TEX_ACCUMULATION), want)
        for n in (2, 3, 8):
            bounds = tiles.balanced_bands(cost, h, n)
            assert len(bounds) == n + 1 and bounds[0] == 0 and bounds[-1] == h and all(a <= b for a, b in zip(bounds, bounds[1:]))
            assert all(b % 8 == 0 for b in bounds[:-1])
            share = [int(cost[bounds[r] // 8:(bounds[r + 1] + 7) // 8].sum()) for r in range(n)]
            assert max(share) <= cost.sum() / n + cost.sum(axis=1).max()      # no band exceeds its share by more than one tile row
            parts = []
            for r in range(n):
                ctx.set_rows(bounds[r], bounds[r + 1] - bounds[r])
                parts.append(_render(ctx, demo, w, h, frames)[0])
            assert pc.same_bits(tiles.stack_bands(parts, bounds), want), bounds
    finally:
        ctx.set_rows(0, -1)
        ctx.resize(64, 64)
