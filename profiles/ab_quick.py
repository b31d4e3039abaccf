#!/usr/bin/env python3
"""A/B of several builds of the library on ONE box, alternating, without touching libmi3pt.so (the build under test is named by
MI3PT_LIBRARY, which the Python host reads).  The scenes are generated once and handed to the child processes as pickles; every
(round, library) is a fresh process.  Per leg: the driver's job shape (80 warm-up frames, then 320 frames in 64-frame launches,
1920x1080, 8 bounces), wall clock around submit .. sync, Mrays/s from the kernel's own ray count.
usage: python profiles/ab_quick.py ROUNDS lib1.so lib2.so ... [--scenes dragon,demo,closeup,forest] [--tile R/N] [--frames 320] [--variants 0,13] [--walk-min N] [--batch FRAMES_PER_LAUNCH]
       [--opt-legs "25=0;25=1"]   (legs of mi3pt_debug_set_option settings, "option=value[,option=value]" each, run for every library)"""
import os, pickle, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
CLOSEUP = {"position": (0.55, 0.62, 1.15), "target": (0.0, 0.5, 0.0)}


def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def child():
    import ptcommon as pc
    from mi3pt_host import capi
    scenes_dir, names = sys.argv[2], sys.argv[3].split(",")
    tile = tuple(int(v) for v in opt("--tile", "0/1").split("/"))
    frames = int(opt("--frames", "320"))
    env = pickle.load(open(os.path.join(scenes_dir, "env.pkl"), "rb"))
    out = []
    for name in names:
        sc = pickle.load(open(os.path.join(scenes_dir, ("dragon" if name == "closeup" else name) + ".pkl"), "rb"))
        if name == "closeup":
            sc.camera = dict(sc.camera, **CLOSEUP)
        W, H = 1920, 1080
        n = frames if name != "forest" else max(frames // 5, 64)
        ctx = capi.Context(0)
        ctx.set_kernel_variant(int(opt("--variant", "0")))
        if "--walk-min" in sys.argv:
            ctx.set_option(capi.OPT_WALK_MIN, int(opt("--walk-min", "0")))
        if "--batch" in sys.argv:                  # frames per launch (MI3PT_OPT_BATCH; before resize)
            ctx.set_option(capi.OPT_BATCH, int(opt("--batch", "64")))
        for kv in [x for x in opt("--opts", "").split(",") if x]:
            ctx.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
        pc.upload_scene(ctx, sc, env)
        ctx.set_tile(tile[0], tile[1], 8)
        ctx.resize(W, H)
        per = ctx.batch_capacity()
        per -= per % 16

        def run(f0, count):
            done = 0
            while done < count:
                k = min(per, count - done)
                ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f0 + done, bounces=8).tobytes())
                ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f0 + done).tobytes())
                ctx.submit_frames(3, k); ctx.flush()
                done += k
        run(2, 80 if name != "forest" else 32)
        ctx.sync(); ctx.reset_counters(); ctx.sync()
        t = time.perf_counter()
        run(100, n)
        ctx.sync()
        dt = time.perf_counter() - t
        c = ctx.counters()
        out.append(f"{name} {c['rays'] / dt / 1e6:7.0f} (v{ctx.active_variant()}, {c['box_tests'] / c['rays']:.1f}+{c['tri_tests'] / c['rays']:.2f})")
        ctx.close()
    print("  ".join(out), flush=True)


def main():
    rounds = int(sys.argv[1])
    libs = [a for a in sys.argv[2:] if a.endswith(".so")]
    names = opt("--scenes", "dragon,demo").split(",")
    from mi3pt_host import scenes
    d = "/tmp/mi3pt_ab_scenes"
    os.makedirs(d, exist_ok=True)
    pickle.dump(scenes.synthetic_env(), open(os.path.join(d, "env.pkl"), "wb"), protocol=4)
    for name in {("dragon" if n == "closeup" else n) for n in names}:
        path = os.path.join(d, name + ".pkl")
        if os.path.exists(path):
            continue
        sc = {"dragon": scenes.dragon_class_scene, "demo": scenes.demo_scene, "forest": scenes.forest_scene}[name]()
        sc.build_bvh()
        pickle.dump(sc, open(path, "wb"), protocol=4)
    extra = [a for k in ("--tile", "--frames", "--walk-min", "--batch") if k in sys.argv for a in (k, opt(k, ""))]
    variants = opt("--variants", "0").split(",")          # several kernel variants of each library, e.g. --variants 0,13
    legs = opt("--opt-legs", "").split(";")
    for r in range(rounds):
        for lib in libs:
            for v in variants:
                for leg in legs:
                    env = dict(os.environ, MI3PT_LIBRARY=os.path.abspath(lib))
                    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", d, ",".join(names), *extra, "--variant", v, *(["--opts", leg] if leg else [])],
                                       env=env, capture_output=True, text=True, timeout=600)
                    line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "FAILED: " + p.stderr[-300:]
                    print(f"{os.path.basename(lib):24s} variant {v:>2s} {leg:12s} {line}", flush=True)


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "--child" else main()
