#!/usr/bin/env python3
"""Experiment (library built with -DPT_DIAG_SERVICE): shader cycles of the parts of a service step.
usage: [WORKLOAD=dragon] python profiles/service_parts.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ptcommon as pc
from mi3pt_host import capi, scenes
sc = scenes.dragon_class_scene() if os.environ.get("WORKLOAD") == "dragon" else scenes.demo_scene()
sc.build_bvh()
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, scenes.synthetic_env())
w, h = 1920, 1080
ctx.resize(w, h)
ctx.enable_wave_times(True)
f = 2
for _ in range(2):
    for _ in range(16):
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f), 3); f += 1
    ctx.sync()
raw = ctx.wave_times(); raw = raw[raw[:, 2] > 0]
hi = lambda x: (x >> np.uint64(32)).astype(np.int64)
service_steps = hi(raw[:, 5]).sum()
node, tri, rest = raw[:, 9].astype(float).sum(), raw[:, 10].astype(float).sum(), raw[:, 11].astype(float).sum()
parts = raw[:, 12:16].astype(float).sum(0)
tot = node + tri + rest + parts.sum()
print(f"node {node / tot:.1%}  triangle {tri / tot:.1%}  service: hit shading {parts[0] / tot:.1%}  miss shading {parts[1] / tot:.1%}  "
      f"refill + camera paths {parts[2] / tot:.1%}  segment starts {parts[3] / tot:.1%}  votes / loop {rest / tot:.1%}")
print("cycles per service step:", " ".join(f"{n} {v / service_steps:.0f}" for n, v in zip(("hit", "miss", "refill+path", "segment", "rest"), list(parts) + [rest])))
