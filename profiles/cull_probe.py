#!/usr/bin/env python3
"""Probe of the distance-culling walks (variant 9: binary packets, 10 - 12: 4-ary wide packets, 13: compressed wide packets) against the reference-counter walk (variant 7):
ms per frame, Mrays/s, box and triangle tests per ray, image identity.
usage: python profiles/cull_probe.py [demo|dragon|forest[:instances]] [WxH] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ptcommon as pc
from mi3pt_host import capi, scenes

what = sys.argv[1] if len(sys.argv) > 1 else "demo"
w, h = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 32
t = time.time()
if what.startswith("forest"):
    sc = scenes.forest_scene(instances=int(what.split(":")[1]) if ":" in what else 9000)
elif what == "dragon":
    sc = scenes.dragon_class_scene()
else:
    sc = scenes.demo_scene()
sc.build_bvh()
print(f"{what}: {len(sc.triangles)} triangles, scene + tree in {time.time() - t:.1f} s", flush=True)
env = scenes.synthetic_env()
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, env)
ctx.resize(w, h)
images = {}
for variant in (7, 9, 10, 11, 12, 13):
    try:
        ctx.set_kernel_variant(variant)
    except capi.Mi3ptError:
        continue                     # (11 / 12: the experiment build only)
    ctx.reset()
    f = 2
    for _ in range(4):
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f), 3); f += 1
    ctx.sync()
    ctx.reset_counters()
    t = time.time()
    for _ in range(nframes):
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f), 3); f += 1
    ctx.sync()
    dt = time.time() - t
    c = ctx.counters()
    images[variant] = ctx.read_texture(capi.TEX_ACCUMULATION)
    print(f"  variant {variant}: {dt / nframes * 1e3:.3f} ms/frame, {c['rays'] / dt / 1e6:.0f} Mrays/s, "
          f"box/ray {c['box_tests'] / c['rays']:.2f}, tri/ray {c['tri_tests'] / c['rays']:.2f}, slow {c['reserved']}", flush=True)
print("  images identical:", all(pc.same_bits(images[7], images[v]) for v in images), " ".join(pc.describe_diff(images[v], images[7]) for v in sorted(images) if v != 7))
