#!/usr/bin/env python3
"""Contiguous, cost-balanced bands (mi3pt_set_rows + mi3pt_measure_tile_cost + tiles.balanced_bands) against the round-robin
8-row tiles (mi3pt_set_tile) for an 8-way split of the headline view: every rank's share rendered alone on this one GPU,
512 frames after a 128-frame warm-up, us per frame -- the slowest rank is the job's time, the sum is what the split costs.
usage: python profiles/probe_bands.py [nranks=8] [floor_per_pixel=0,20,40]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ptcommon as pc
from mi3pt_host import capi, scenes, tiles
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
floors = [float(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,20,40").split(",")]
sc = scenes.dragon_class_scene(); sc.build_bvh(); env = scenes.synthetic_env()
W, H, FRAMES, WARM = 1920, 1080, 512, 128
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, env)


def job(frames, f0):
    per = ctx.batch_capacity()
    done = 0
    while done < frames:
        k = min(per, frames - done)
        ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f0 + done, bounces=8).tobytes())
        ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f0 + done).tobytes())
        ctx.submit_frames(3, k); ctx.flush()
        done += k


def timed():
    if ctx.local_rows == 0:
        return 0.0, 0
    job(WARM, 2); ctx.sync(); ctx.reset_counters(); ctx.sync()
    t = time.perf_counter(); job(FRAMES, 1000); ctx.sync()
    return (time.perf_counter() - t) / FRAMES * 1e6, ctx.counters()["rays"] // FRAMES


ctx.resize(W, H)
whole_us, whole_rays = timed()
print(f"whole image: {whole_us:8.2f} us/frame, {whole_rays} rays/frame, {whole_rays / whole_us:.0f} Mrays/s", flush=True)
ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=2, bounces=8).tobytes())
t = time.perf_counter(); cost = ctx.measure_tile_cost(); dt = time.perf_counter() - t
print(f"measuring frame: {dt * 1e3:.2f} ms; cost per tile row (x1000): " + " ".join(f"{int(v) // 1000}" for v in cost.sum(axis=1)[::9]), flush=True)
us = []
for r in range(N):
    ctx.set_tile(r, N, 8); ctx.resize(W, H)
    us.append(timed()[0])
print(f"round-robin 8-row tiles: " + " ".join(f"{v:6.2f}" for v in us) + f"  | slowest {max(us):6.2f}  sum {sum(us):7.2f}  ideal {whole_us / N:6.2f} us/frame", flush=True)
ctx.set_tile(0, 1, 8)
for fl in floors:
    b = tiles.balanced_bands(cost, H, N, floor_per_pixel=fl)
    us = []
    for r in range(N):
        ctx.set_rows(b[r], b[r + 1] - b[r]); ctx.resize(W, H)
        us.append(timed()[0])
    print(f"balanced bands (floor {fl:g}/pixel) {b}: " + " ".join(f"{v:6.2f}" for v in us) + f"  | slowest {max(us):6.2f}  sum {sum(us):7.2f}", flush=True)
    # a second cut from the MEASURED times of the first (what a job that repeats could do): rows re-dealt in proportion
    if fl == floors[-1]:
        rate = [u / max(b[r + 1] - b[r], 1) for r, u in enumerate(us)]      # us per row in each band
        row_us = np.concatenate([np.full(b[r + 1] - b[r], rate[r]) for r in range(N)])
        tr = np.add.reduceat(row_us, np.arange(0, H, 8))
        b2 = tiles.balanced_bands((tr[:, None] * 1000).astype(np.int64), H, N)
        us2 = []
        for r in range(N):
            ctx.set_rows(b2[r], b2[r + 1] - b2[r]); ctx.resize(W, H)
            us2.append(timed()[0])
        print(f"re-cut from measured times {b2}: " + " ".join(f"{v:6.2f}" for v in us2) + f"  | slowest {max(us2):6.2f}  sum {sum(us2):7.2f}", flush=True)
ctx.close()
