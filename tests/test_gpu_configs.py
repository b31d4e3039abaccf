"""BASELINE.json configs 2-5 as parity-test cases at the sizes BASELINE.json states (config 1 is in
test_gpu_parity.py / test_golden.py).  The shipped kernel is held to the ORACLE on whole images:
config 2 for its whole 64-spp job, config 3 -- the headline's -- for its whole 256-spp job, config 4
for two whole 4K frames with the thin lens (and four row blocks for all 1024 spp), config 5 for a quarter of the 4K image (two ranks of the 8-way split; a whole
frame of the 10 M-triangle forest is most of a minute of oracle).  The oracle's passes run on
the box's CPU share (pt_oracle.default_threads: the cgroup quota, not the 256 processors OpenMP
sees).  Beside that: bit-identity with the per-pixel kernel (variant 2, the WGSL control flow),
counter identities, and tile-split reassembly (the multi-GPU decomposition run rank by rank on
the one GPU)."""
import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi, tiles, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dragon(built):
    sc = scenes.dragon_class_scene()          # 869,882 triangles
    sc.build_bvh()
    return sc


@pytest.fixture(scope="module")
def forest(built):
    sc = scenes.forest_scene()                # 9000 instances = 9,972,002 triangles (full size)
    sc.build_bvh()
    return sc


def _render(ctx, sc, w, h, frames, variant=0, **kw):
    ctx.set_kernel_variant(variant)
    ctx.reset()
    ctx.reset_counters()
    for f in frames:
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8, **kw), pc.acc_uniforms(w, h, f),
                     capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
    img = ctx.read_texture(capi.TEX_ACCUMULATION)
    cnt = ctx.counters()
    ctx.set_kernel_variant(0)
    return img, cnt


def _oracle_image(orc, sc, env, w, h, frames, rank=0, nranks=1, rows=8, **kw):
    """The oracle's running mean over `frames` of rank `rank`'s share of an `nranks`-way split in `rows`-row blocks
    (rank 0 of 1 = the whole image), and its counters summed over the frames."""
    osc = pc.oracle_scene(orc, sc, env)
    acc = np.zeros((orc.tile_local_rows(h, rank, nranks, rows), w, 4), np.float32)
    total = {}
    for f in frames:
        part, cnt = orc.raytrace(osc, pc.rt_uniforms(sc, w, h, frame=f, bounces=8, **kw).tobytes(), w, h, rank, nranks, rows)
        acc = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, part, acc, rank, nranks, rows)
        for k, v in cnt.items():
            total[k] = total.get(k, 0) + v
    return acc, total


def _oracle_block(orc, sc, env, w, h, frames, first_row, rows=8, **kw):
    """One `rows`-row block of the image starting at first_row (a multiple of rows), over `frames`."""
    nblocks = (h + rows - 1) // rows
    assert first_row % rows == 0 and first_row < h
    return _oracle_image(orc, sc, env, w, h, frames, first_row // rows, nblocks, rows, **kw)[0]


def _same_as_oracle(got, cgot, want, cwant, what):
    assert pc.same_bits(got, want), what + ": " + pc.describe_diff(got, want)
    assert pc.max_rel_err(got, want) <= 1e-4          # north_star's bar (implied by the line above)
    pc.check_counters(cgot, cwant, culled=True, what=what)


def test_config3_dragon_class_1080p(gpu_ctx, orc, dragon, env):
    w, h = 1920, 1080
    ctx = gpu_ctx
    pc.upload_scene(ctx, dragon, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    frames = (2, 3, 4)
    ref, cref = _render(ctx, dragon, w, h, frames, variant=2)
    got, cgot = _render(ctx, dragon, w, h, frames, variant=0)
    assert pc.same_bits(got, ref), pc.describe_diff(got, ref)
    pc.check_counters(cgot, cref, culled=True, what="shipped kernel vs per-pixel kernel")
    same, csame = _render(ctx, dragon, w, h, frames, variant=7)        # the walk that runs exactly the reference's tests
    assert pc.same_bits(same, ref)
    pc.check_counters(csame, cref, culled=False, what="deferred-leaf kernel vs per-pixel kernel")
    assert cgot["box_tests"] < 0.8 * cref["box_tests"]
    assert cgot["pixels"] == 3 * w * h and cgot["rays"] == cgot["hits"] + cgot["misses"]
    assert cgot["stack_overflows"] == 0 and cgot["reserved"] < 0.02 * cgot["rays"]
    assert np.isfinite(got).all()
    want, cwant = _oracle_image(orc, dragon, env, w, h, frames)          # the WHOLE image, three frames
    _same_as_oracle(got, cgot, want, cwant, "config 3, shipped kernel vs oracle")
    pc.check_counters(cref, cwant, culled=False, what="per-pixel kernel vs oracle")


def test_config4_dragon_dof_denoise_4k_tile_split(gpu_ctx, orc, dragon, env):
    """Thin-lens DoF (aperture 0.03, focus at the model) at 3840x2160; 4-rank tile split
    reassembles to the whole image; the de-noise + ACES pass runs on the gathered image."""
    w, h = 3840, 2160
    ctx = gpu_ctx
    pc.upload_scene(ctx, dragon, env)
    focal = float(np.linalg.norm(np.array(dragon.camera["position"]) - np.array([0.0, 0.5, 0.0])))
    kw = dict(aperture=0.03, focal=focal)
    frames = (2, 3)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    whole, cwhole = _render(ctx, dragon, w, h, frames, **kw)
    f = pc.fs_uniforms(w, h, 1.0, 1, 1)
    ctx.set_uniforms(capi.PASS_FULLSCREEN, f.tobytes())
    ctx.submit(capi.SUBMIT_FULLSCREEN)
    canvas = ctx.read_canvas_rgba8()
    assert canvas.shape == (h, w, 4) and (canvas[..., 3] == 255).all() and canvas[..., :3].std() > 5
    out = np.zeros_like(whole)
    rays = 0
    for rank in range(4):
        ctx.set_tile(rank, 4, 8)
        ctx.resize(w, h)
        part, c = _render(ctx, dragon, w, h, frames, **kw)
        rows = tiles.local_rows_of(h, rank, 4, 8)
        out[rows] = part
        rays += c["rays"]
    assert pc.same_bits(out, whole), pc.describe_diff(out, whole)
    assert rays == cwhole["rays"]
    want, cwant = _oracle_image(orc, dragon, env, w, h, frames, **kw)    # the WHOLE 4K image with the lens, two frames
    _same_as_oracle(whole, cwhole, want, cwant, "config 4, shipped kernel vs oracle")
    # the de-noise pass against the oracle on a crop (the filter only looks 6 texels around)
    crop = whole[960:1060, 1800:2000].copy()
    fc = pc.fs_uniforms(200, 100, 1.0, 1, 1)
    want, _ = orc.fullscreen(fc.tobytes(), crop)
    ctx.set_tile(0, 1, 8)
    ctx.resize(200, 100)
    ctx.write_texture(capi.TEX_ACCUMULATION, crop)      # hand the image to this (1-rank) context
    ctx.set_uniforms(capi.PASS_FULLSCREEN, fc.tobytes())
    ctx.submit(capi.SUBMIT_FULLSCREEN)
    got = ctx.read_texture(capi.TEX_CANVAS)
    inner = (slice(8, -8), slice(8, -8))         # away from the crop's wrap-around border
    assert pc.same_bits(got[inner], want[inner]), pc.describe_diff(got[inner], want[inner])
    ctx.set_tile(0, 1, 8)
    ctx.resize(64, 64)


def test_config5_forest_10m_triangles_4k_tile_split_8(gpu_ctx, orc, forest, env):
    """BASELINE.json config 5 at FULL size: the 10 M-triangle instanced forest at 3840x2160, 8
    bounces.  The reference cannot bind this scene (128 MiB storage-buffer limit, renderer.ts:512:
    1.12 GB of triangles), so there is nothing to compare with but the build's own references:
    shipped kernel == deferred-leaf kernel (the reference's exact box / triangle tests) == per-pixel
    kernel, bit for bit, counters consistent; 8-way tile split, rank by rank, == whole image; two
    ranks' shares (a quarter of the image, blocks from top to bottom) == the oracle."""
    w, h = 3840, 2160
    ctx = gpu_ctx
    assert len(forest.triangles) > 9_900_000
    pc.upload_scene(ctx, forest, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    frames = (2,)
    ref, cref = _render(ctx, forest, w, h, frames, variant=2)          # per-pixel kernel, the WGSL control flow
    same, csame = _render(ctx, forest, w, h, frames, variant=7)        # persistent kernel, same tests
    whole, cwhole = _render(ctx, forest, w, h, frames)                 # shipped: + distance culling
    assert pc.same_bits(same, ref), pc.describe_diff(same, ref)
    assert pc.same_bits(whole, ref), pc.describe_diff(whole, ref)
    pc.check_counters(csame, cref, culled=False, what="deferred-leaf kernel vs per-pixel kernel")
    pc.check_counters(cwhole, cref, culled=True, what="shipped kernel vs per-pixel kernel")
    wide8, cwide8 = _render(ctx, forest, w, h, frames, variant=14)     # the eight-wide walk (an option): its packets exist for this tree, same bits
    assert ctx.last_launch()["variant"] == 14
    assert pc.same_bits(wide8, ref), pc.describe_diff(wide8, ref)
    pc.check_counters(cwide8, cref, culled=True, what="eight-wide walk vs per-pixel kernel")
    assert cref["pixels"] == w * h and cref["stack_overflows"] == 0
    assert cref["box_tests"] > 300 * cref["rays"]          # the reference walk: hundreds of boxes per ray in this scene
    assert np.isfinite(whole).all()
    out = np.zeros_like(whole)
    rays = 0
    for rank in range(8):
        ctx.set_tile(rank, 8, 8)
        ctx.resize(w, h)
        part, c = _render(ctx, forest, w, h, frames)
        out[tiles.local_rows_of(h, rank, 8, 8)] = part
        rays += c["rays"]
        if rank in (2, 5):          # these ranks' shares against the oracle: 2 x 1/8 of the 4K image
            want, cwant = _oracle_image(orc, forest, env, w, h, frames, rank, 8, 8)
            _same_as_oracle(part, c, want, cwant, f"config 5, rank {rank} of 8 vs oracle")
    assert pc.same_bits(out, whole), pc.describe_diff(out, whole)
    assert rays == cwhole["rays"]
    ctx.set_tile(0, 1, 8)
    ctx.resize(64, 64)


# ---------------------------------------------------------------- the configs at their stated sample counts

def _render_spp(ctx, sc, w, h, spp, variant=0, **kw):
    """`spp` consecutive 1-spp frames (renderer.ts:369-377: frame = 2, 3, ...) into the running mean."""
    ctx.set_kernel_variant(variant)
    ctx.reset()
    ctx.reset_counters()
    ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, w, h, frame=2, bounces=8, **kw).tobytes())
    ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
    ctx.submit_frames(capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE, spp)
    img = ctx.read_texture(capi.TEX_ACCUMULATION)
    cnt = ctx.counters()
    ctx.set_kernel_variant(0)
    return img, cnt


def _same_job(got, cgot, ref, cref, pixels):
    assert pc.same_bits(got, ref), pc.describe_diff(got, ref)
    for k in pc.PATH_COUNTERS:
        assert cgot[k] == cref[k], (k, cgot[k], cref[k])
    assert cgot["pixels"] == pixels and cgot["rays"] == cgot["hits"] + cgot["misses"] and cgot["stack_overflows"] == 0
    assert (got[..., 3] == 1.0).all()          # (a NaN colour is legitimate: normalize() of a zero vector once in ~10^8 paths, as in the reference)


def test_config2_demo_1080p_64spp(gpu_ctx, orc, demo, env):
    """BASELINE.json config 2 at its stated sample count: 64 spp = one 64-frame launch of the shipped
    kernel against the ORACLE's 64 whole frames and their running mean, and against 64 fused
    per-pixel-kernel frames (the WGSL control flow verbatim): the same image, bit for bit."""
    w, h, spp = 1920, 1080, 64
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    ref, cref = _render_spp(ctx, demo, w, h, spp, variant=2)
    got, cgot = _render_spp(ctx, demo, w, h, spp)
    assert ctx.active_variant() == 13          # the shipped choice: compressed wide packets (round 4) wherever the tree admits them
    _same_job(got, cgot, ref, cref, spp * w * h)
    want, cwant = _oracle_image(orc, demo, env, w, h, range(2, 2 + spp))
    _same_as_oracle(got, cgot, want, cwant, "config 2, 64 spp, shipped kernel vs oracle")
    for v in pc.variants_available(ctx, (10, 11, 12)):                     # (10: what `auto` picked among the exact-packet walks for this scene -- short walks, thin floor leaves)
        a, ca = _render_spp(ctx, demo, w, h, spp, variant=v)
        _same_job(a, ca, ref, cref, spp * w * h)
    ctx.resize(64, 64)


def test_config3_dragon_class_1080p_256spp(gpu_ctx, orc, dragon, env):
    """Config 3 -- the configuration the headline metric is quoted on -- at its stated 256 spp: one 256-frame launch of the shipped
    kernel (the default batch depth, the six-wave build) against the ORACLE's 256 whole frames and their running mean (about 75 s of
    oracle on the box's 16-core share), and against the per-pixel kernel."""
    w, h, spp = 1920, 1080, 256
    ctx = gpu_ctx
    pc.upload_scene(ctx, dragon, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    ref, cref = _render_spp(ctx, dragon, w, h, spp, variant=2)
    got, cgot = _render_spp(ctx, dragon, w, h, spp)
    assert ctx.active_variant() == 13          # the shipped choice: compressed wide packets, one-axis culling condition (this scene's hint)
    _same_job(got, cgot, ref, cref, spp * w * h)
    assert cgot["tri_tests"] <= cref["tri_tests"]
    assert ctx.last_launch()["workgroups"] == 24 * 256          # six waves per SIMD: 8.3 M jobs in the launch
    want, cwant = _oracle_image(orc, dragon, env, w, h, range(2, 2 + spp))
    _same_as_oracle(got, cgot, want, cwant, "config 3, 256 spp, the WHOLE image vs oracle")
    # the eight-wide walk (variant 14: round 6's experiment, an option) at the same 256 spp: the same image, the same paths
    w8, cw8 = _render_spp(ctx, dragon, w, h, spp, variant=14)
    assert ctx.last_launch()["variant"] == 14 and ctx.last_launch()["lean"]
    _same_as_oracle(w8, cw8, want, cwant, "config 3, 256 spp, the eight-wide walk vs oracle")
    for v in pc.variants_available(ctx, (10, 11, 12)):                     # ... and the exact-packet wide walks, on a shorter job
        a, ca = _render_spp(ctx, dragon, w, h, 16, variant=v)
        b, cb = _render_spp(ctx, dragon, w, h, 16)
        _same_job(a, ca, b, cb, 16 * w * h)
    ctx.resize(64, 64)


def test_walk_threshold_follows_the_view(built, dragon, env):
    """MI3PT_OPT_WALK_ADAPT (round 6): a launch of the shipped walk reports what it cost per ray, and later launches run the deep-walk build
    (walk_min 44: made for very large trees) while the view tests more than 40 boxes per ray -- the 870 k-triangle mesh from close up: 68 --
    and the ordinary one again below 30 (its stated view: 17).  Which build runs never changes a bit: every image equals the per-pixel
    kernel's.  Own context: the test changes options."""
    w, h, per = 960, 544, 32                 # 120 x 68 tiles x 32 frames = 261 k jobs per launch: above the 250 k a deep launch needs
    mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
    close = np.array([0.55, 0.62, 1.15]), np.array([0.0, 0.5, 0.0])
    views = {"close": dict(position=tuple(close[0]), direction=tuple((close[1] - close[0]) / np.linalg.norm(close[1] - close[0]))), "stated": {}}
    with capi.Context(0) as ctx:
        pc.upload_scene(ctx, dragon, env)
        ctx.set_option(capi.OPT_BATCH, per)
        ctx.resize(w, h)
        assert ctx.get_option(capi.OPT_WALK_ADAPT) == 1

        def launch(view, frame0, variant=0):
            ctx.set_kernel_variant(variant)
            ctx.reset()
            ctx.reset_counters()
            ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(dragon, w, h, frame=frame0, bounces=8, **views[view]).tobytes())
            ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, frame0).tobytes())
            ctx.submit_frames(mask, per)
            img = ctx.read_texture(capi.TEX_ACCUMULATION)          # (waits: the launch's report has arrived when this returns)
            c = ctx.counters()
            return img, c["box_tests"] / max(c["rays"], 1), ctx.last_launch()

        seen = []
        for view, frame0 in (("close", 2), ("close", 40), ("close", 80), ("stated", 2), ("stated", 40), ("close", 120)):
            img, per_ray, last = launch(view, frame0)
            ref, _, _ = launch(view, frame0, variant=2)
            assert pc.same_bits(img, ref), f"{view} from frame {frame0}: " + pc.describe_diff(img, ref)
            assert last["variant"] == 13 and last["lean"]
            seen.append((view, round(per_ray), last["walk_min"]))
        # the first close-up launch runs the ordinary build and reports ~68 boxes per ray; the next ones run the deep build; the first launch of
        # the stated view still does (nothing has reported on it yet), reports ~17, and the ordinary build is back; then the close-up again
        assert [s[2] for s in seen] == [32, 44, 44, 44, 32, 32], seen
        assert seen[0][1] > 40 and seen[3][1] < 30, seen
        ctx.set_option(capi.OPT_WALK_ADAPT, 0)
        for frame0 in (2, 40):
            _, _, last = launch("close", frame0)
            assert last["walk_min"] == 32
        ctx.set_kernel_variant(0)


def test_config4_dragon_dof_4k_1024spp_one_rank_of_4(gpu_ctx, orc, dragon, env):
    """Config 4 at its stated 1024 spp, for one rank of the 4-way tile split (the ranks share nothing
    but the final gather; the reassembly is tested above): thin lens on, 512-frame launches; one
    8-row block through the model against the oracle's 1024 frames."""
    w, h, spp = 3840, 2160, 1024
    ctx = gpu_ctx
    pc.upload_scene(ctx, dragon, env)
    focal = float(np.linalg.norm(np.array(dragon.camera["position"]) - np.array([0.0, 0.5, 0.0])))
    kw = dict(aperture=0.03, focal=focal)
    ctx.set_tile(1, 4, 8)
    ctx.resize(w, h)
    assert ctx.batch_capacity() == 512
    ref, cref = _render_spp(ctx, dragon, w, h, spp, variant=2, **kw)
    got, cgot = _render_spp(ctx, dragon, w, h, spp, **kw)
    _same_job(got, cgot, ref, cref, spp * w * ctx.local_rows)
    mine = tiles.local_rows_of(h, 1, 4, 8)
    for lo in (400, 1100, 1400, 1900):                                 # four blocks of this rank's: sky, through the model (twice), floor
        first = next(y for y in mine if y >= lo and y % 8 == 0)
        want = _oracle_block(orc, dragon, env, w, h, range(2, 2 + spp), first, **kw)
        at = mine.index(first)
        assert pc.same_bits(got[at:at + 8], want), f"config 4, 1024 spp, the block at row {first} vs oracle: " + pc.describe_diff(got[at:at + 8], want)
    ctx.set_tile(0, 1, 8)
    ctx.resize(64, 64)


def test_config5_forest_4k_4096spp_one_rank_of_8(gpu_ctx, orc, forest, env):
    """Config 5 at its stated 4096 spp, for one rank of the 8-way split: sixteen 256-frame launches of
    the shipped walk (4-ary packets, filtered slab test) against the binary culling walk; the per-pixel kernel, which
    the one-frame test above holds both to, would take a minute for this many frames.  The oracle (0.4 Mrays/s in this
    scene) holds one block of the rank's to its first 8 frames -- the same launch path, a shorter job."""
    w, h, spp = 3840, 2160, 4096
    ctx = gpu_ctx
    pc.upload_scene(ctx, forest, env)
    ctx.set_tile(3, 8, 8)
    ctx.resize(w, h)
    ref, cref = _render_spp(ctx, forest, w, h, spp, variant=9)
    got, cgot = _render_spp(ctx, forest, w, h, spp)
    assert ctx.active_variant() == 13          # the shipped choice: compressed wide packets, three-axis culling condition (this scene's margins are not negligible)
    _same_job(got, cgot, ref, cref, spp * w * ctx.local_rows)
    short, _ = _render_spp(ctx, forest, w, h, 32)
    mine = tiles.local_rows_of(h, 3, 8, 8)
    first = next(y for y in mine if y >= 1200 and y % 8 == 0)          # a block of this rank's among the trees
    want = _oracle_block(orc, forest, env, w, h, range(2, 34), first)
    at = mine.index(first)
    assert pc.same_bits(short[at:at + 8], want), "config 5, 32 spp, one block vs oracle: " + pc.describe_diff(short[at:at + 8], want)
    ctx.set_tile(0, 1, 8)
    ctx.resize(64, 64)
