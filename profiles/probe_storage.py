#!/usr/bin/env python3
"""Texel storage formats on the bench scene: MI3PT_STORAGE_F32 (the default here) against MI3PT_STORAGE_F16 (the reference's
rgba16float, renderer.ts:102): ms per 320-frame job (dragon-class, 1080p, 8 bounces), best of 5, and which kernel ran."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptcommon as pc
from mi3pt_host import capi, scenes
sc = scenes.dragon_class_scene(); sc.build_bvh(); env = scenes.synthetic_env()
W, H = 1920, 1080
for storage, name in ((capi.STORAGE_F32, "f32"), (capi.STORAGE_F16, "f16"), (capi.STORAGE_F32, "f32"), (capi.STORAGE_F16, "f16")):
    ctx = capi.Context(0)
    ctx.set_storage(storage)
    pc.upload_scene(ctx, sc, env)
    ctx.resize(W, H)
    f, times = 2, []
    for rep in range(6):
        ctx.sync(); ctx.reset_counters(); t = time.perf_counter()
        for _ in range(5):
            ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f, bounces=8).tobytes())
            ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f).tobytes())
            ctx.submit_frames(3, 64); ctx.flush(); f += 64
        ctx.sync(); times.append((time.perf_counter() - t) * 1e3)
    rays = ctx.counters()["rays"]
    print(f"storage {name}: 320-frame jobs best {min(times[1:]):.3f} ms = {rays / min(times[1:]) / 1e3:.0f} Mrays/s (variant {ctx.active_variant()})", flush=True)
    ctx.close()
