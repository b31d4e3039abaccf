#!/usr/bin/env python3
"""Where the deep-walk build starts to pay: the 870 k-triangle scene from cameras between its stated view and the close-up, the ordinary build
(walk_min 32) against the deep one (44), forced; box tests per ray beside the speeds.  The thresholds of MI3PT_OPT_WALK_ADAPT (40 / 30) come from here.
usage: python profiles/probe_walk_view.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ptcommon as pc
from mi3pt_host import capi, scenes

sc = scenes.dragon_class_scene(); sc.build_bvh()
env = scenes.synthetic_env()
W, H, PER = 1920, 1080, 160
p0, t0 = np.array(sc.camera["position"], float), np.array(sc.camera.get("target", (0.0, 0.0, 0.0)), float)
p1, t1 = np.array([0.55, 0.62, 1.15]), np.array([0.0, 0.5, 0.0])
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, env)
ctx.set_option(capi.OPT_BATCH, PER)
ctx.resize(W, H)
for s in (0.0, 0.2, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0):
    pos, tgt = p0 + s * (p1 - p0), t0 + s * (t1 - t0)
    d = (tgt - pos) / np.linalg.norm(tgt - pos)
    out = []
    for wm in (32, 44, 32, 44):
        ctx.set_option(capi.OPT_WALK_MIN, wm)
        def run(f0, n):
            for k in range(n):
                ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f0 + k * PER, bounces=8, position=tuple(pos), direction=tuple(d)).tobytes())
                ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f0 + k * PER).tobytes())
                ctx.submit_frames(3, PER); ctx.flush()
        run(2, 1); ctx.sync(); ctx.reset_counters(); ctx.sync()
        t = time.perf_counter(); run(500, 2); ctx.sync(); dt = time.perf_counter() - t
        c = ctx.counters()
        out.append((wm, c["rays"] / dt / 1e6, c["box_tests"] / c["rays"]))
    n32 = np.mean([o[1] for o in out if o[0] == 32]); n44 = np.mean([o[1] for o in out if o[0] == 44])
    print(f"camera {s:.1f} of the way to the close-up: {out[0][2]:5.1f} boxes per ray   ordinary {n32:7.0f}   deep {n44:7.0f} Mrays/s   deep / ordinary {n44 / n32:.3f}", flush=True)
ctx.close()
