#!/usr/bin/env python3
"""Times the fullscreen pass (de-noise + ACES + RGBA8) on an accumulated demo image.
usage: python profiles/fullscreen_time.py [WxH ...]   (needs a GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import ptcommon as pc  # noqa: E402
from mi3pt_host import capi, scenes  # noqa: E402

sizes = sys.argv[1:] or ["1920x1080", "3840x2160"]
sc = scenes.demo_scene()
sc.build_bvh()
env = scenes.synthetic_env()
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, env)
ctx.enable_timing(True)
for size in sizes:
    w, h = (int(v) for v in size.split("x"))
    ctx.resize(w, h)
    for f in range(2, 6):
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f), 3)
    for scaling in (1.0, 0.25):
        for denoise in (1, 0):
            ctx.set_uniforms(capi.PASS_FULLSCREEN, pc.fs_uniforms(w, h, scaling, denoise, 1).tobytes())
            times = []
            for _ in range(6):
                ctx.submit(capi.SUBMIT_FULLSCREEN)
                ctx.sync()
                times.append(ctx.pass_time_us(capi.PASS_FULLSCREEN))
            print(f"{size} scaling {scaling} denoise {denoise}: fullscreen pass {np.median(times[1:]):.1f} us")
