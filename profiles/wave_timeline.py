#!/usr/bin/env python3
"""Diagnostic: per-wave lifetime of the persistent raytrace kernel (demo scene, 8 bounces).

usage: python profiles/wave_timeline.py [WxH]
Prints when the resident waves begin, see the work queue run empty, and end (100 MHz wall
clock), plus the shader clock they averaged.  Needs a GPU.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import ptcommon as pc  # noqa: E402
from mi3pt_host import capi, scenes  # noqa: E402

w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
sc = scenes.demo_scene()
sc.build_bvh()
env = scenes.synthetic_env()
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, env)
ctx.resize(w, h)
ctx.set_kernel_variant(4)
ctx.enable_wave_times(True)
for frame in (2, 3, 4):
    pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=frame, bounces=8), pc.acc_uniforms(w, h, frame), 3)
ctx.sync()
t = ctx.wave_times().astype(np.int64)
t = t[t[:, 2] > 0]
b, e, end, clk = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
t0 = b.min()


def us(x):
    return (x - t0) / 100.0


print(f"{w}x{h}: waves {len(t)}  kernel span {us(end.max()):.1f} us")
print(f"begin    min/median/max us: {us(b).min():.1f} {np.median(us(b)):.1f} {us(b).max():.1f}")
ee = e[e > 0]
if len(ee):
    print(f"empty    min/median/max us: {us(ee).min():.1f} {np.median(us(ee)):.1f} {us(ee).max():.1f}")
print(f"end      min/median/max us: {us(end).min():.1f} {np.median(us(end)):.1f} {us(end).max():.1f}")
life = (end - b) / 100.0
print(f"lifetime min/median/max us: {life.min():.1f} {np.median(life):.1f} {life.max():.1f}")
print(f"shader clock over lifetime (GHz): median {np.median(clk / (life * 1e3)):.3f}")
hist, edges = np.histogram(us(end), bins=10)
print("end-time histogram:", [(float(a.round(0)), int(n)) for a, n in zip(edges[:-1], hist)])
hist, edges = np.histogram(us(b), bins=10)
print("begin-time histogram:", [(float(a.round(0)), int(n)) for a, n in zip(edges[:-1], hist)])
