#!/usr/bin/env python3
"""Small raytrace launches (an interactive host: one or a few frames per launch) against the number of persistent waves
per CU: ms per frame for launches of B frames, queued back to back (q) and with a sync after each (s).
usage: python profiles/probe_small_launches.py [demo|dragon|blob:<segments>] [WxH]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import ptcommon as pc  # noqa: E402
from mi3pt_host import capi, scenes  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "demo"
w, h = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
if workload == "demo":
    sc = scenes.demo_scene()
    sc.build_bvh()
elif workload.startswith("blob:"):          # the dragon-class generator at another size (2 x segments^2 triangles)
    sc = scenes.dragon_class_scene(segments=int(workload[5:]))
    sc.build_bvh()
else:
    import bench
    sc, _ = bench.build_scene("dragon")
env = scenes.synthetic_env()
RT_ACC = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
print(f"{workload} ({len(sc.triangles)} triangles) {w}x{h}: ms per frame, queued / sync after every launch")
for batch in (1, 2, 4, 8):
    row = []
    for waves in (0, 20, 16, 12, 10, 8, 6, 4):
        ctx = capi.Context(0)
        if waves:
            ctx.set_option(capi.OPT_WAVES_PER_CU, waves)
        ctx.set_option(capi.OPT_BATCH, batch)
        pc.upload_scene(ctx, sc, env)
        ctx.resize(w, h)
        frames = 48
        res = []
        for sync_each in (False, True):
            def loop(first):
                for f in range(first, first + frames):
                    pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f), RT_ACC)
                    if sync_each and (f - first + 1) % batch == 0:
                        ctx.sync()
                ctx.sync()
            loop(2)
            t0 = time.perf_counter()
            loop(2 + frames)
            res.append((time.perf_counter() - t0) / frames * 1e3)
        row.append(f"{'auto' if not waves else waves}: {res[0]:.3f}/{res[1]:.3f}")
        ctx.close()
    print(f"  {batch} frame(s) per launch   " + "   ".join(row), flush=True)
