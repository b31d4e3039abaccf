#!/usr/bin/env python3
"""Probe of BASELINE.json configs 3-5 on one GPU: scene build times, frame times, the
fullscreen (de-noise + ACES) pass at 4K.  usage: python profiles/configs_probe.py [forest_instances]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ptcommon as pc
from mi3pt_host import capi, scenes

def frames(ctx, sc, w, h, n, first=2, **kw):
    for f in range(first, first + n):
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8, **kw), pc.acc_uniforms(w, h, f), 3)

env = scenes.synthetic_env()
ctx = capi.Context(0)
ctx.enable_timing(True)
# config 3/4: dragon-class, 4K, DoF + de-noise
t = time.time(); sc = scenes.dragon_class_scene(); t_gen = time.time() - t
t = time.time(); sc.build_bvh(); t_bvh = time.time() - t
print(f"dragon-class: {len(sc.triangles)} tris, gen {t_gen:.2f}s, bvh {t_bvh:.2f}s")
pc.upload_scene(ctx, sc, env)
for (w, h) in ((1920, 1080), (3840, 2160)):
    ctx.resize(w, h)
    frames(ctx, sc, w, h, 4, aperture=0.03, focal=4.1); ctx.sync(); ctx.reset_counters()
    t = time.time(); frames(ctx, sc, w, h, 32, first=6, aperture=0.03, focal=4.1); ctx.sync(); dt = time.time() - t
    c = ctx.counters()
    print(f"  {w}x{h} DoF: {dt / 32 * 1e3:.3f} ms/frame, {c['rays'] / dt / 1e6:.0f} Mrays/s, box/ray {c['box_tests'] / c['rays']:.1f}")
    ctx.set_uniforms(capi.PASS_FULLSCREEN, pc.fs_uniforms(w, h, 1.0, 1, 1).tobytes())
    ctx.submit(capi.SUBMIT_FULLSCREEN); ctx.sync()
    t = time.time(); ctx.submit(capi.SUBMIT_FULLSCREEN); ctx.sync(); print(f"  fullscreen de-noise+ACES: {(time.time() - t) * 1e3:.2f} ms (GPU {ctx.pass_time_us(2) / 1e3:.2f} ms)")
# config 5: forest
inst = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
t = time.time(); sc = scenes.forest_scene(instances=inst); t_gen = time.time() - t
t = time.time(); sc.build_bvh(); t_bvh = time.time() - t
n = sc.nodes; internal = n["isLeaf"] != 1
print(f"forest: {len(sc.triangles)} tris, gen {t_gen:.1f}s, bvh {t_bvh:.1f}s, scene bytes {(sc.triangles.nbytes + n.nbytes) / 1e9:.2f} GB")
t = time.time(); pc.upload_scene(ctx, sc, env); print(f"  upload {time.time() - t:.1f}s")
w, h = 3840, 2160
ctx.resize(w, h)
frames(ctx, sc, w, h, 2); ctx.sync(); ctx.reset_counters()
t = time.time(); frames(ctx, sc, w, h, 16, first=4); ctx.sync(); dt = time.time() - t
c = ctx.counters()
print(f"  {w}x{h}: {dt / 16 * 1e3:.3f} ms/frame, {c['rays'] / dt / 1e6:.0f} Mrays/s, box/ray {c['box_tests'] / c['rays']:.1f}, tri/ray {c['tri_tests'] / c['rays']:.2f}, overflows {c['stack_overflows']}")
