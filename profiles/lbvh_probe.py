#!/usr/bin/env python3
"""Device-built linear BVH vs the reference's SAH tree: build time and rendering speed.
usage: python profiles/lbvh_probe.py [forest_instances]   (needs a GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import ptcommon as pc  # noqa: E402
from mi3pt_host import capi, scenes  # noqa: E402

env = scenes.synthetic_env()
ctx = capi.Context(0)
w, h = 1920, 1080


def frames(sc, n, first=2):
    for f in range(first, first + n):
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f), 3)


def measure(sc, nodes, label):
    t = time.time()
    ctx.upload_bvh(nodes)
    t_up = time.time() - t
    ctx.resize(w, h)
    frames(sc, 16)
    ctx.sync()
    ctx.reset_counters()
    t = time.time()
    frames(sc, 32, first=18)
    ctx.sync()
    dt = time.time() - t
    c = ctx.counters()
    img = ctx.read_texture(capi.TEX_ACCUMULATION)
    print(f"  {label}: upload+analysis {t_up:.2f} s, {dt / 32 * 1e3:.3f} ms/frame, {c['rays'] / dt / 1e6:.0f} Mrays/s, "
          f"{c['box_tests'] / c['rays']:.1f} box + {c['tri_tests'] / c['rays']:.2f} triangle tests per ray, stack aborts {c['stack_overflows']}")
    return img


inst = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
for name, make in (("demo", scenes.demo_scene), ("dragon-class", scenes.dragon_class_scene), (f"forest x{inst}", lambda: scenes.forest_scene(instances=inst))):
    sc = make()
    ctx.upload_triangles(sc.triangles)
    ctx.upload_materials(sc.material_bytes)
    ctx.upload_environment(env)
    t = time.time(); sah = sc.build_bvh(); t_sah = time.time() - t
    t = time.time(); lb, ms = ctx.device_build_bvh(); t_lb = time.time() - t
    print(f"{name}: {len(sc.triangles)} triangles; SAH build on the host {t_sah:.2f} s; device LBVH {ms:.2f} ms on the GPU ({t_lb:.2f} s incl. read-back)")
    a = measure(sc, sah, "reference (SAH) tree")
    b = measure(sc, lb, "device LBVH tree   ")
    diff = int((a != b).any(-1).sum())
    print(f"  pixels that differ between the two trees after 48 frames: {diff} of {w * h}")
