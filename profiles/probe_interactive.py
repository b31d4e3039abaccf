#!/usr/bin/env python3
"""A host that looks at every frame (default scene, 1920x1080, 8 bounces): ms per frame for
  a) one raytrace launch per frame, nothing waited for until the end (MI3PT_OPT_BATCH 1)
  b) the same with a sync after every frame
  c) raytrace + accumulate + fullscreen per frame and a sync after every frame (what an interactive viewer does)
  d) c + the 8-bit canvas read back every frame
Per leg also: the GPU time of the raytrace launches alone (event pairs) -- what is left is launch latency, the other passes and the
host's round trip.  usage: [WORKLOAD=dragon] python profiles/probe_interactive.py [frames [waves per CU]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptcommon as pc  # noqa: E402
from mi3pt_host import capi, scenes  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
waves_per_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # 0: the library's choice
w, h = 1920, 1080
sc = scenes.dragon_class_scene() if os.environ.get("WORKLOAD") == "dragon" else scenes.demo_scene()
sc.build_bvh()
ctx = capi.Context(0)
if waves_per_cu:
    ctx.set_option(capi.OPT_WAVES_PER_CU, waves_per_cu)
    print("waves per CU:", waves_per_cu)
pc.upload_scene(ctx, sc, scenes.synthetic_env())
ctx.resize(w, h)
ctx.enable_timing(True)
if os.environ.get("COST_ORDER"):          # MI3PT_OPT_COST_ORDER: 1 = the cheapest quarter of the tiles last, 2 = all tiles costliest first
    ctx.set_option(capi.OPT_COST_ORDER, int(os.environ["COST_ORDER"]))
    print("cost order:", os.environ["COST_ORDER"])
ctx.set_uniforms(capi.PASS_FULLSCREEN, pc.fs_uniforms(w, h, 1.0, 1, 1).tobytes())
RT_ACC = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE


def loop(mask, sync_each, read_each, first):
    for f in range(first, first + frames):
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f), mask)
        if read_each:
            ctx.read_canvas_rgba8()
        elif sync_each:
            ctx.sync()
    ctx.sync()


for name, batch, mask, sync_each, read_each in (
        ("a) a launch per frame, queued", 1, RT_ACC, False, False),
        ("b) a launch per frame, sync per frame", 1, RT_ACC, True, False),
        ("c) + fullscreen pass, sync per frame", 64, RT_ACC | capi.SUBMIT_FULLSCREEN, True, False),
        ("d) + canvas read back per frame", 64, RT_ACC | capi.SUBMIT_FULLSCREEN, False, True)):
    ctx.set_option(capi.OPT_BATCH, batch)
    loop(mask, sync_each, read_each, 2)
    ctx.reset_counters()
    ctx.raytrace_launch_stats(reset=True)
    t0 = time.perf_counter()
    loop(mask, sync_each, read_each, 2 + frames)
    dt = time.perf_counter() - t0
    ms, launches, nf = ctx.raytrace_launch_stats()
    print(f"{name:42s} {dt / frames * 1e3:7.4f} ms per frame  {ctx.counters()['rays'] / dt / 1e6:8.1f} Mrays/s   raytrace launches: {launches} of {nf / max(launches, 1):.1f} frames, "
          f"{ms / max(launches, 1):.4f} ms each on the GPU clock; waves {ctx.last_launch()['workgroups']}")
