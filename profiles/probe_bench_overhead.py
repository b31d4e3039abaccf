#!/usr/bin/env python3
"""What the first job after a pause costs: one rank of an 8-way split renders the driver's 320-frame job (a) repeatedly,
back to back, (b) after host-side pauses of 1 / 5 / 20 ms with the GPU idle, (c) after a pause during which a tiny
kernel keeps the GPU awake.  ms per job.  (Round 3: the first job after bench.py's warm-up took 12.7 ms, a repeated
one 11.7.)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from mi3pt_host import capi
capi.load_library()
torch.cuda.set_device(0)
job = bench.Job("dragon", 1920, 1080, 3, 8, 0, 0, None)
ctx = job.ctx
per_launch = bench.frames_per_launch(ctx.batch_capacity())
job.frames(80, per_launch); ctx.sync()


def one():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job.frames(320, per_launch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


print("first after warm-up: %.3f ms" % one())
print("back to back: " + ", ".join("%.3f" % one() for _ in range(4)))
for pause in (0.001, 0.005, 0.020, 0.100):
    vals = []
    for _ in range(3):
        time.sleep(pause)
        vals.append(one())
    print("after %5.1f ms idle: %s" % (pause * 1e3, ", ".join("%.3f" % v for v in vals)))
x = torch.zeros(1 << 20, device="cuda")
for pause in (0.005, 0.020):
    vals = []
    for _ in range(3):
        t_end = time.perf_counter() + pause
        while time.perf_counter() < t_end:
            x.add_(1.0)
        vals.append(one())
    print("after %5.1f ms of small kernels: %s" % (pause * 1e3, ", ".join("%.3f" % v for v in vals)))
ctx.close()
