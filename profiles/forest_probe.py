#!/usr/bin/env python3
"""Diagnostic: ms per frame of batched launches on the 10 M-triangle forest at 3840x2160 (one rank of an 8-way split),
for launch depths 16 / 64 / 256, with the GPU time of the launches themselves.  usage: python profiles/forest_probe.py"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'webgpu-pathtracer_amd', 'py')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import ptcommon as pc
from mi3pt_host import capi, scenes
sc = scenes.forest_scene(); sc.build_bvh()
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, scenes.synthetic_env())
w, h = 3840, 2160
r, n = (int(x) for x in os.environ.get('TILE', '3/8').split('/'))
ctx.set_tile(r, n, 8); ctx.resize(w, h)
ctx.enable_timing(True)
for spp in tuple(int(x) for x in os.environ.get('SPP', '256,16,64,256').split(',')):
    ctx.reset(); ctx.reset_counters(); ctx.raytrace_launch_stats(reset=True)
    ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, w, h, frame=2, bounces=8).tobytes())
    ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, 2).tobytes())
    ctx.sync(); t0 = time.perf_counter()
    ctx.submit_frames(3, spp); ctx.sync()
    dt = time.perf_counter() - t0
    ms, n, f = ctx.raytrace_launch_stats()
    print(f"spp {spp}: wall {dt*1e3/spp:.3f} ms/frame; {n} launches, {f} frames, kernel {ms/max(f,1):.3f} ms/frame", flush=True)
