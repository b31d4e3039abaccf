#!/usr/bin/env python3
"""How a 320-frame job of one rank of an 8-way split (the driver's --steps 20) is best cut into launches:
256 + 64 (the batch limit of round 2), 160 + 160, or one launch of 320 (batch limit 512).  ms per job, best of 5."""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptcommon as pc
from mi3pt_host import capi, scenes
sc = scenes.dragon_class_scene(); sc.build_bvh(); env = scenes.synthetic_env()
W, H = 1920, 1080
for tile in ((3, 8), (1, 4), (0, 1)):
    for limit, cuts in ((256, (256, 64)), (256, (160, 160)), (512, (320,)), (512, (512, 512, 256)), (256, (256,) * 5), (1024, (1280,))):
        ctx = capi.Context(0)
        ctx.set_option(capi.OPT_BATCH_LIMIT, limit)
        ctx.set_option(capi.OPT_BATCH, max(64, limit // tile[1]))
        pc.upload_scene(ctx, sc, env)
        ctx.set_tile(tile[0], tile[1], 8); ctx.resize(W, H)
        f = 2
        best = 1e9
        for rep in range(6):
            ctx.sync(); t = time.perf_counter()
            for n in cuts:
                ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f, bounces=8).tobytes())
                ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f).tobytes())
                ctx.submit_frames(3, n); ctx.flush(); f += n
            ctx.sync(); dt = (time.perf_counter() - t) * 1e3
            if rep: best = min(best, dt)
        print(f"tile {tile[0]}/{tile[1]} limit {limit} cuts {cuts}: {best:.3f} ms for {sum(cuts)} frames = {best / sum(cuts) * 1e3:.2f} us/frame (capacity {ctx.batch_capacity()})", flush=True)
        ctx.close()
