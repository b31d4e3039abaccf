#!/usr/bin/env python3
"""Diagnostic: per-wave lifetime and step statistics of the persistent raytrace kernel.

usage: [WORKLOAD=dragon|forest] [TILE=R/N] [VARIANT=v] [LITE=1] python profiles/wave_timeline.py [WxH] [frames_per_launch]
LITE=1 (experiment build of the library, MI3PT_LIBRARY=.../libmi3pt_exp.so): the LEAN kernel with lane counts per kind of step --
the binary that ships plus a dozen scalar counters, five waves per SIMD -- instead of the four-wave diagnostic twin; it has no
per-step clock stamps (no cycle split, no drain statistics).  The kernel's Mrays/s with and without the counters is printed.
Renders one launch of `frames_per_launch` batched frames (default 16; 1 = a single frame) of
the demo (or dragon-class) scene at 8 bounces and prints when the resident waves begin, see the
work queue run empty, and end (100 MHz wall clock), the shader clock they averaged, and how
well filled the walk / service steps were.  Needs a GPU.
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
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc = {"dragon": scenes.dragon_class_scene, "forest": scenes.forest_scene}.get(os.environ.get("WORKLOAD"), scenes.demo_scene)()
sc.build_bvh()
env = scenes.synthetic_env()
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, env)
if os.environ.get("TILE"):               # TILE=R/N: rank R's share of an N-way tile split
    ctx.set_tile(*(int(v) for v in os.environ["TILE"].split("/")), 8)
ctx.resize(w, h)
if os.environ.get("VARIANT"):
    ctx.set_kernel_variant(int(os.environ["VARIANT"]))
lite = os.environ.get("LITE") == "1"
frame = 2


def batches(count):
    """`count` launches of nframes frames; (ms per launch of the last, rays of the last)"""
    global frame
    import time
    for _ in range(count):
        ctx.sync(); ctx.reset_counters(); ctx.sync()
        t = time.perf_counter()
        ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, w, h, frame=frame, bounces=8).tobytes())
        ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(w, h, frame).tobytes())
        ctx.submit_frames(3, nframes); ctx.flush(); ctx.sync()
        dt = time.perf_counter() - t
        frame += nframes
    return dt * 1e3, ctx.counters()["rays"]


plain_ms, plain_rays = batches(3)        # the shipped kernel, nothing bound
plain_launch = ctx.last_launch()
ctx.enable_wave_times(True)
if lite:
    ctx.set_option(capi.OPT_DIAG_LITE, 1)
diag_ms, diag_rays = batches(2)          # warm-up batch, then the measured one
diag_launch = ctx.last_launch()
print(f"kernel under the counters: variant {diag_launch['variant']} {'lean + lane counts (LITE)' if diag_launch['lean'] else 'diagnostic twin'}, "
      f"{diag_launch['workgroups']} waves, {diag_rays / diag_ms / 1e3:.0f} Mrays/s;  shipped kernel, same launches: variant {plain_launch['variant']}, "
      f"{plain_launch['workgroups']} waves, {plain_rays / plain_ms / 1e3:.0f} Mrays/s  ({(diag_rays / diag_ms) / (plain_rays / plain_ms) - 1:+.1%})")
raw = ctx.wave_times()
raw = raw[raw[:, 2] > 0]
t = raw[:, :4].astype(np.int64)
b, e, end, clk = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
t0 = b.min()


def us(x):
    return (x - t0) / 100.0


print(f"{w}x{h} x {nframes} frames per launch: waves {len(t)}  kernel span {us(end.max()):.1f} us "
      f"({us(end.max()) / nframes:.1f} us per frame)")
print(f"begin    min/median/max us: {us(b).min():.1f} {np.median(us(b)):.1f} {us(b).max():.1f}")
ee = e[e > 0]
if len(ee):
    print(f"empty    min/median/max us: {us(ee).min():.1f} {np.median(us(ee)):.1f} {us(ee).max():.1f}")
print(f"end      min/median/max us: {us(end).min():.1f} {np.median(us(end)):.1f} {us(end).max():.1f}")
life = (end - b) / 100.0
print(f"lifetime min/median/max us: {life.min():.1f} {np.median(life):.1f} {life.max():.1f}")
print(f"shader clock over lifetime (GHz): median {np.median(clk / (life * 1e3)):.3f}")


def hi(x):
    return (x >> np.uint64(32)).astype(np.int64)


def lo(x):
    return (x & np.uint64(0xFFFFFFFF)).astype(np.int64)


walk_steps, walk_lanes = hi(raw[:, 4]).sum(), lo(raw[:, 4]).sum()
service_steps, leaf_lanes = hi(raw[:, 5]).sum(), lo(raw[:, 5]).sum()
shade_lanes, hit_lanes = hi(raw[:, 6]).sum(), lo(raw[:, 6]).sum()
path_lanes, segment_lanes = hi(raw[:, 7]).sum(), lo(raw[:, 7]).sum()
print(f"walk steps {walk_steps} ({walk_steps / nframes / 1e6:.3f} M per frame), mean walking lanes/step "
      f"{walk_lanes / max(walk_steps, 1):.1f}" + ("" if lo(raw[:, 8]).any() else f", of which on a leaf {leaf_lanes / max(walk_steps, 1):.1f}"))
hit_steps, b_steps = hi(raw[:, 12]).sum(), lo(raw[:, 12]).sum()       # service steps that served the hit group / the miss + path group
print(f"service steps {service_steps} ({service_steps / nframes / 1e3:.1f} k per frame): {hit_steps} shaded hits, "
      f"{hit_lanes / max(hit_steps, 1):.1f} lanes each; {b_steps} shaded misses / started paths, "
      f"{shade_lanes / max(b_steps, 1):.1f} miss lanes + {path_lanes / max(b_steps, 1):.1f} path starts each; "
      f"segment starts/service step {segment_lanes / max(service_steps, 1):.1f}")
tri_steps = lo(raw[:, 8]).sum() if raw.shape[1] > 8 else 0
parked = hi(raw[:, 8]).sum() if raw.shape[1] > 8 else 0       # triangles parked in the wave, summed over its triangle steps
if tri_steps:
    print(f"deferred-leaf walk: the walk steps above are node steps; triangle steps {tri_steps} "
          f"({tri_steps / nframes / 1e6:.3f} M per frame), lanes/triangle step {leaf_lanes / tri_steps:.1f}; triangles parked in the "
          f"wave when a triangle step starts {parked / tri_steps:.1f} ({parked / max(leaf_lanes, 1):.2f} per lane that has any)")
print(f"steps per wave: walk {walk_steps / len(raw):.0f}  triangle {tri_steps / len(raw):.0f}  service {service_steps / len(raw):.0f}")
if raw.shape[1] > 11 and raw[:, 9:12].any():
    cyc = raw[:, 9:12].astype(np.float64).sum(0)
    tot = cyc.sum()
    print(f"shader cycles by kind of step (deferred-leaf walk): node {cyc[0] / tot:.1%}  triangle {cyc[1] / tot:.1%}  "
          f"service (+ loop overhead) {cyc[2] / tot:.1%};  per step: node {cyc[0] / max(walk_steps, 1):.0f}  "
          f"triangle {cyc[1] / max(tri_steps, 1):.0f}  service {cyc[2] / max(service_steps, 1):.0f} cycles")


if raw.shape[1] > 15 and raw[:, 13:16].any():
    tn, tt = hi(raw[:, 13]), lo(raw[:, 13])
    ts, live = hi(raw[:, 14]), lo(raw[:, 14])
    tl = raw[:, 15].astype(np.int64)
    tail_us = (end - e)[e > 0] / 100.0
    steps = (tn + tt + ts)
    print(f"after the queue ran empty, per wave (median / max): {np.median(tail_us):.0f} / {tail_us.max():.0f} us, "
          f"live lanes at that moment {np.median(live):.0f} / {live.max()}, steps node {np.median(tn):.0f} / {tn.max()}  "
          f"triangle {np.median(tt):.0f} / {tt.max()}  service {np.median(ts):.0f} / {ts.max()}, "
          f"lanes per node-or-triangle step {tl.sum() / max((tn + tt).sum(), 1):.1f}")
    slow = np.argsort(end)[-5:]
    for i in slow:
        print(f"  a late wave: empty at {us(e[i]):.0f} us, end {us(end[i]):.0f} us, live {live[i]}, tail steps node {tn[i]} triangle {tt[i]} service {ts[i]}")
