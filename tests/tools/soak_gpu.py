#!/usr/bin/env python3
"""Randomised differential soaks on a GPU box -- longer relatives of the randomised tests in tests/test_gpu_parity.py, run
by hand (DESIGN.md "Randomised differential tests" quotes a run of each).  Every soak compares bits and stops after a few
failures, printing the offending configuration.

    python tests/tools/soak_gpu.py [configs] [presentation] [fullscreen] [groups] [options] [cameras] [oracle]   (default: all)

  configs       the shipped batched path against the per-pixel kernel over random sizes / bounces / spp / lens / tile
                splits / storage / batch cuts, on three scenes (1 200 configurations)
  presentation  exact presentation with shared launches against a launch per presenting frame, random call sequences
  fullscreen    the fullscreen pass against the oracle over random sizes, scalings, resolution uniforms, textures
  groups        device groups of 1 .. 6 members (all on device 0) against one context
  options       random settings of all scheduling options at once against the defaults
  cameras       random cameras (inside the model, straight down, zero direction components), shipped path against the
                per-pixel kernel
  oracle        the same kind of cameras against the CPU oracle (small images)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ("tests", os.path.join("webgpu-pathtracer_amd", "py"), "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np  # noqa: E402
import ptcommon as pc  # noqa: E402
from mi3pt_host import capi, layout, scenes  # noqa: E402

MASK = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
ENV = scenes.synthetic_env()


def scene(name):
    sc = scenes.demo_scene() if name == "demo" else scenes.dragon_class_scene(segments=int(name[4:]))
    sc.build_bvh()
    return sc


def report(name, n, fails, t0):
    print(f"{name}: {n} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
    return fails


def soak_configs():
    import test_gpu_parity as T
    t0, n, fails = time.time(), 0, 0
    for name in ("demo", "blob30", "blob120"):
        sc = scene(name)
        with capi.Context(0) as ctx:
            for seed in range(12, 412):
                try:
                    T.test_random_configurations_shipped_path_equals_per_pixel_kernel(ctx, sc, ENV, seed)
                except AssertionError as e:
                    fails += 1
                    print("FAIL", name, seed, str(e)[:300], flush=True)
                n += 1
                if fails > 3:
                    break
    return report("configs", n, fails, t0)


def soak_presentation():
    import test_gpu_parity as T
    t0, n, fails = time.time(), 0, 0
    sc = scene("demo")
    with capi.Context(0) as ctx:              # (reused on purpose: a test's shared context arrives in any state)
        for seed in range(100, 180):
            try:
                T.test_presentation_differential_over_random_call_sequences(ctx, sc, ENV, seed)
            except AssertionError as e:
                fails += 1
                print("FAIL", seed, str(e)[:200], flush=True)
            n += 1
    return report("presentation", n, fails, t0)


def soak_fullscreen(cases=400):
    import pt_oracle as orc
    rng = np.random.default_rng(7)
    t0, fails = time.time(), 0
    with capi.Context(0) as ctx:
        for _ in range(cases):
            w, h = int(rng.integers(1, 140)), int(rng.integers(1, 100))
            scaling = float(rng.choice([1.0, 1.0, 0.93, 1.07, 0.5, 0.25, 2.0, float(rng.random() * 1.5 + 0.01)]))
            rs = float(rng.choice([1.0, 1.0, 0.9, 1.1, 0.37, 3.0, float(rng.random() * 2 + 0.05)]))
            dn, tm = int(rng.random() < 0.85), int(rng.integers(0, 3))
            ctx.set_tile(0, 1, 8)
            ctx.resize(w, h)
            kind = rng.integers(0, 3)
            tex = rng.random((h, w, 4), dtype=np.float32) * np.float32(2.0 if kind else 0.1)
            if kind == 2:
                tex = np.float32(0.3) + tex * np.float32(0.02)
            tex[..., 3] = 1.0
            ctx.write_texture(capi.TEX_ACCUMULATION, tex)
            f = layout.UniformBlock(layout.FULLSCREEN_UNIFORMS)
            f.set({"resolution": [w * rs, h * rs], "aspect": w / h, "scalingFactor": scaling, "denoise": dn, "tonemapping": tm})
            ctx.set_uniforms(capi.PASS_FULLSCREEN, f.tobytes())
            ctx.submit(capi.SUBMIT_FULLSCREEN)
            got, got8 = ctx.read_texture(capi.TEX_CANVAS), ctx.read_canvas_rgba8()
            want, want8 = orc.fullscreen(f.tobytes(), tex)
            if not (pc.same_bits(got, want) and np.array_equal(got8, want8)):
                fails += 1
                print("FAIL", w, h, scaling, rs, dn, tm, pc.describe_diff(got, want)[:200], flush=True)
                if fails > 5:
                    break
    return report("fullscreen", cases, fails, t0)


def soak_groups(cases=60):
    import test_gpu_group as G
    rng = np.random.default_rng(11)
    t0, fails = time.time(), 0
    sc = scene("demo")
    with capi.Context(0) as single:
        for _ in range(cases):
            members, w, h = int(rng.integers(1, 7)), int(rng.integers(1, 200)), int(rng.integers(1, 160))
            frames, per_call, rows = int(rng.integers(1, 20)), int(rng.choice([1, 2, 5, 24])), int(rng.choice([8, 8, 4, 16]))
            single.set_tile(0, 1, 8)
            want = G._job(single, sc, ENV, w, h, frames, per_call)
            with capi.Context(devices=[0] * members, block_rows=rows) as g:
                got = G._job(g, sc, ENV, w, h, frames, per_call)
            ok = all(a.shape == b.shape and pc.same_bits(a.astype(np.float32), b.astype(np.float32)) for a, b in zip(got[:3], want[:3]))
            ok = ok and all(got[3][k] == want[3][k] for k in pc.PATH_COUNTERS)
            if not ok:
                fails += 1
                print("FAIL", members, w, h, frames, per_call, rows, flush=True)
    return report("groups", cases, fails, t0)


def render(ctx, sc, w, h, frames, **kw):
    ctx.resize(w, h)
    ctx.reset_counters()
    for f in range(2, 2 + frames):
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, **kw), pc.acc_uniforms(w, h, f), MASK)
    return ctx.read_texture(capi.TEX_ACCUMULATION), ctx.counters()


def soak_options():
    rng = np.random.default_rng(5)
    t0, n, fails = time.time(), 0, 0
    for name in ("demo", "blob60"):
        sc = scene(name)
        with capi.Context(0) as base:
            pc.upload_scene(base, sc, ENV)
            for _ in range(int(os.environ.get("SOAK_OPTIONS", "40"))):
                w, h, frames = int(rng.choice([64, 200, 333, 640])), int(rng.choice([40, 113, 360])), int(rng.choice([1, 3, 16, 40]))
                want, cw = render(base, sc, w, h, frames, bounces=6)
                opts = {capi.OPT_WALK_MIN: int(rng.choice([1, 8, 24, 32, 48, 64])), capi.OPT_LEAF_MIN: int(rng.choice([1, 8, 24, 32, 64])),
                        capi.OPT_SHADE_SPLIT: int(rng.choice([1, 16, 48, 64])), capi.OPT_TAIL_POLICY: int(rng.integers(0, 8)),
                        capi.OPT_TRI_PAIR: int(rng.integers(0, 2)), capi.OPT_JOB_REVERSE: int(rng.integers(0, 2)),
                        capi.OPT_JOB_GROUP: int(rng.choice([-1, 0, 1, 7, 100, 5000])), capi.OPT_JOB_CHUNK: int(rng.choice([1, 2, 4, 7, 64])),
                        capi.OPT_BATCH: int(rng.choice([1, 2, 5, 16, 64])), capi.OPT_WAVES_PER_CU: int(rng.choice([0, 1, 3, 8, 12, 16, 20, 24])),
                        capi.OPT_GATE: int(rng.integers(0, 2)), capi.OPT_COST_ORDER: int(rng.integers(0, 2)),
                        capi.OPT_WIDE: int(rng.integers(0, 2)), capi.OPT_CULL: int(rng.integers(0, 2)),
                        # round 5: five / six waves per SIMD (walk_min 44 picks the deep builds: six waves = a 25-entry stack and the parked
                        # path state in memory), the camera base image, the packet numbering
                        capi.OPT_SIX_WAVES: int(rng.integers(-1, 2)), capi.OPT_CAMERA_BASE: int(rng.integers(0, 2)),
                        capi.OPT_PACKET_ORDER: int(rng.integers(0, 3)),
                        # round 6: how the tree's nodes are grouped into packets (greedy / SAH-optimal / by the packet width)
                        capi.OPT_COLLAPSE: int(rng.integers(-1, 2))}
                if rng.random() < 0.3:
                    opts[capi.OPT_WALK_MIN] = 44
                variant = 14 if rng.random() < 0.4 else 0          # round 6: the eight-wide walk (an option) in four cases of ten
                with capi.Context(0) as c:
                    for k, v in opts.items():
                        c.set_option(k, v)
                    c.set_kernel_variant(variant)
                    pc.upload_scene(c, sc, ENV)
                    got, cg = render(c, sc, w, h, frames, bounces=6)
                n += 1
                if not (pc.same_bits(got, want) and all(cg[k] == cw[k] for k in pc.PATH_COUNTERS)):
                    fails += 1
                    print("FAIL", name, w, h, frames, "variant", variant, opts, flush=True)
    return report("options", n, fails, t0)


def random_camera(rng):
    pos = (rng.random(3) * 6 - 3).tolist()
    if rng.random() < 0.3:
        pos = [float(rng.random() - 0.5), float(rng.random()), float(rng.random() - 0.5)]      # inside / on the model
    d = rng.standard_normal(3)
    if rng.random() < 0.35:
        d[int(rng.integers(0, 3))] = 0.0                                                        # a component of exactly zero
    if rng.random() < 0.1:
        d = np.array([0.0, -1.0, 0.0])
    d = (d / (np.linalg.norm(d) or 1.0)).tolist()
    return dict(bounces=int(rng.choice([1, 3, 8])), position=pos, direction=d, fov=float(rng.choice([20.0, 45.0, 120.0])),
                aperture=float(rng.choice([0.0, 0.1])), focal=float(rng.choice([0.5, 3.0])))


def soak_cameras():
    rng = np.random.default_rng(21)
    t0, n, fails = time.time(), 0, 0
    for name in ("demo", "blob90"):
        sc = scene(name)
        with capi.Context(0) as ctx:
            pc.upload_scene(ctx, sc, ENV)
            for _ in range(150):
                w, h, frames, kw = int(rng.choice([33, 96, 200])), int(rng.choice([17, 64, 120])), int(rng.choice([1, 4, 9])), random_camera(rng)
                res = []
                for variant, pipelined in ((0, True), (1, False), (14, True)):          # (round 6: + the eight-wide walk)
                    ctx.set_kernel_variant(variant)
                    ctx.set_pipelining(pipelined)
                    res.append(render(ctx, sc, w, h, frames, **kw))
                (a, ca), (b, cb), (c8, cc8) = res
                n += 1
                if not (pc.same_bits(a, b) and pc.same_bits(c8, b) and all(ca[k] == cb[k] == cc8[k] for k in pc.PATH_COUNTERS)):
                    fails += 1
                    print("FAIL", name, w, h, kw, pc.describe_diff(a, b)[:200], flush=True)
    return report("cameras", n, fails, t0)


def soak_oracle():
    import pt_oracle as orc
    rng = np.random.default_rng(33)
    t0, n, fails = time.time(), 0, 0
    for name in ("demo", "blob40"):
        sc = scene(name)
        osc = pc.oracle_scene(orc, sc, ENV)
        with capi.Context(0) as ctx:
            pc.upload_scene(ctx, sc, ENV)
            for _ in range(40):
                w, h, kw = int(rng.choice([33, 48])), int(rng.choice([17, 40])), random_camera(rng)
                got, _ = render(ctx, sc, w, h, 2, **kw)
                acc = np.zeros((h, w, 4), np.float32)
                for f in (2, 3):
                    img, _ = orc.raytrace(osc, pc.rt_uniforms(sc, w, h, frame=f, **kw).tobytes(), w, h)
                    acc = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, img, acc)
                n += 1
                if not pc.same_bits(got, acc):
                    fails += 1
                    print("FAIL", name, w, h, kw, pc.describe_diff(got, acc)[:200], flush=True)
    return report("oracle", n, fails, t0)


def main():
    soaks = {"configs": soak_configs, "presentation": soak_presentation, "fullscreen": soak_fullscreen, "groups": soak_groups,
             "options": soak_options, "cameras": soak_cameras, "oracle": soak_oracle}
    chosen = sys.argv[1:] or list(soaks)
    total = sum(soaks[name]() for name in chosen)
    print("soak:", "no difference" if total == 0 else f"{total} failures")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
