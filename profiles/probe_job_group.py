#!/usr/bin/env python3
"""Band size of the job order (MI3PT_OPT_JOB_GROUP: tiles per band; every frame of a band before the next band): ms per
320-frame job, dragon-class 1080p, whole image and one rank of 8, best of 4 repeats."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptcommon as pc
from mi3pt_host import capi, scenes
sc = scenes.dragon_class_scene(); sc.build_bvh(); env = scenes.synthetic_env()
W, H = 1920, 1080
tiles_x = W // 8
for tile in ((0, 1), (3, 8)):
    rows = (H // tile[1] + 7) // 8
    for group_rows in (-1, 1, 2, 4, 8, 16, 32, 64, 0):
        ctx = capi.Context(0)
        ctx.set_option(capi.OPT_JOB_GROUP, group_rows if group_rows <= 0 else group_rows * tiles_x)
        pc.upload_scene(ctx, sc, env)
        ctx.set_tile(tile[0], tile[1], 8); ctx.resize(W, H)
        f, times = 2, []
        for rep in range(5):
            ctx.sync(); t = time.perf_counter(); done = 0
            while done < 320:
                n = min(ctx.batch_capacity(), 320 - done)
                ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f, bounces=8).tobytes())
                ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f).tobytes())
                ctx.submit_frames(3, n); ctx.flush(); f += n; done += n
            ctx.sync(); times.append((time.perf_counter() - t) * 1e3)
        label = {-1: "default (1/8 of the tile rows)", 0: "frame-major"}.get(group_rows, f"{group_rows} tile rows")
        print(f"tile {tile[0]}/{tile[1]} ({rows} tile rows) band = {label}: best {min(times[1:]):.3f} ms", flush=True)
        ctx.close()
