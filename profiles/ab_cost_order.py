#!/usr/bin/env python3
"""Cost-ordered jobs on / off (MI3PT_OPT_COST_ORDER): ms per 320-frame job of one rank of an 8-way split and of the whole image,
best of 5 repeated jobs and the wave timeline's drain figures.  usage: python profiles/ab_cost_order.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptcommon as pc
from mi3pt_host import capi, scenes
sc = scenes.dragon_class_scene(); sc.build_bvh(); env = scenes.synthetic_env()
W, H = 1920, 1080
for tile in ((3, 8), (1, 4), (0, 1)):
    for order in (0, 1, 0, 1):
        ctx = capi.Context(0)
        ctx.set_option(capi.OPT_COST_ORDER, order)
        pc.upload_scene(ctx, sc, env)
        ctx.set_tile(tile[0], tile[1], 8); ctx.resize(W, H)
        f, times = 2, []
        for rep in range(7):
            ctx.sync(); t = time.perf_counter()
            done = 0
            while done < 320:
                n = min(ctx.batch_capacity(), 320 - done)
                ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f, bounces=8).tobytes())
                ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f).tobytes())
                ctx.submit_frames(3, n); ctx.flush(); f += n; done += n
            ctx.sync(); times.append((time.perf_counter() - t) * 1e3)
        print(f"tile {tile[0]}/{tile[1]} cost order {order}: 320-frame jobs {', '.join('%.2f' % x for x in times)} ms; best of the last five {min(times[2:]):.3f}", flush=True)
        ctx.close()
