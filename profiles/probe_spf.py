#!/usr/bin/env python3
"""samplesPerFrame 1 / 2 / 4 / 16 (the reference's slider, main.ts:188), pipelining off (a launch per frame) and maxBounces 0 on the
shipped kernel: Mrays/s on the dragon-class scene at 1080p, the same number of samples per pixel in every leg (64), deep
launches.  Round 3 sent everything but samplesPerFrame == 1 to a 128-VGPR twin with scratch; round 4 keeps the per-pixel sum
and the sample count in the pixel's texel of the frame's radiance slot, so every setting runs the lean 96-VGPR build.
usage: python profiles/probe_spf.py [demo|dragon] [samples|frames]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptcommon as pc
from mi3pt_host import capi, scenes
name = (sys.argv[1:] or ["dragon"])[0]
sc = scenes.demo_scene() if name == "demo" else scenes.dragon_class_scene()
sc.build_bvh(); env = scenes.synthetic_env()
W, H, SPP = 1920, 1080, 64
MODE = (sys.argv[2:] or ["samples"])[0]        # "samples": 64 samples per pixel in every leg; "frames": 64 frames in every leg (the reference's `frames` setting stays, main.ts:181)
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, env)
ctx.resize(W, H)


def job(spf, bounces=8, pipelined=True, reps=4):
    ctx.set_pipelining(pipelined)
    frames = SPP // spf if MODE == "samples" else SPP
    best, rays = 1e9, 0
    f = 2
    for rep in range(reps):
        ctx.reset(); ctx.reset_counters(); ctx.sync()
        t = time.perf_counter()
        ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f, bounces=bounces, spf=spf).tobytes())
        ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f).tobytes())
        ctx.submit_frames(3, frames); ctx.flush(); ctx.sync()
        dt = time.perf_counter() - t
        if rep:
            best = min(best, dt)
        rays = ctx.counters()["rays"]
        f += frames
    ctx.set_pipelining(True)
    return rays / best / 1e6, best * 1e3, frames


print(f"{name} {W}x{H}, " + (f"{SPP} samples per pixel" if MODE == "samples" else f"{SPP} frames") + " per leg, kernel variant {ctx.active_variant() if hasattr(ctx, 'active_variant') else '?'}")
for spf in (1, 2, 4, 16):
    r, ms, frames = job(spf)
    print(f"samplesPerFrame {spf:2d}: {frames:3d} frames, {ms:8.3f} ms, {r:8.0f} Mrays/s", flush=True)
r, ms, frames = job(1, pipelined=False)
print(f"samplesPerFrame  1, pipelining off (a raytrace + an accumulate launch per frame): {frames} frames, {ms:8.3f} ms, {r:8.0f} Mrays/s", flush=True)
r, ms, frames = job(16, pipelined=False)
print(f"samplesPerFrame 16, pipelining off: {frames} frames, {ms:8.3f} ms, {r:8.0f} Mrays/s", flush=True)
_, ms, frames = job(1, bounces=0)
print(f"maxBounces 0: {frames} frames, {ms:8.3f} ms ({ms / frames * 1e3:.1f} us per frame of {W * H} black pixels)", flush=True)
ctx.close()
