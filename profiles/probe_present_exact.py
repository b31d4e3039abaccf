#!/usr/bin/env python3
"""The exact-canvas loop (MI3PT_PRESENT_EXACT: raytrace + accumulate + fullscreen per frame, the reference's
renderer.ts:366-395) on the default scene at 1920x1080, 8 bounces: ms per frame and Mrays/s; with `trace <csv>` the
kernel timeline of a rocprofv3 --kernel-trace run of this script is printed (start, duration, what ran beside it).
usage: python profiles/probe_present_exact.py [frames [present depth]]
       rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 profiles/probe_present_exact.py 32
       python profiles/probe_present_exact.py trace DIR/.../*_kernel_trace.csv"""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def trace(path, last=40):
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    ks = ks[-last:]
    t0 = ks[0][0]
    prev_end = {}
    for s, e, name in ks:
        short = name.split("(")[0].split("::")[-1][:28]
        beside = [n for (s2, e2, n) in ks if n != name and s2 < e and e2 > s]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {short:28s} beside: {', '.join(sorted(set(b.split('(')[0].split('::')[-1][:16] for b in beside)))}")
    span = ks[-1][1] - ks[0][0]
    print(f"{len(ks)} kernels in {span / 1e3:.1f} us")


def main():
    import ptcommon as pc
    from mi3pt_host import capi, scenes
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    w, h = 1920, 1080
    sc = scenes.demo_scene()
    sc.build_bvh()
    ctx = capi.Context(0)
    pc.upload_scene(ctx, sc, scenes.synthetic_env())
    ctx.resize(w, h)
    ctx.set_present_mode(capi.PRESENT_EXACT)
    if depth:
        ctx.set_option(capi.OPT_PRESENT_DEPTH, depth)
    ctx.set_uniforms(capi.PASS_FULLSCREEN, pc.fs_uniforms(w, h, 1.0, 1, 1).tobytes())
    everything = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE | capi.SUBMIT_FULLSCREEN

    def run(first):
        for f in range(first, first + frames):
            pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=f, bounces=8), pc.acc_uniforms(w, h, f), everything)
        ctx.sync()

    run(2)
    ctx.reset_counters()
    t0 = time.perf_counter()
    run(2 + frames)
    dt = time.perf_counter() - t0
    rays = ctx.counters()["rays"]
    print(f"present_exact (depth {depth or 'default'}): {frames} frames, {dt / frames * 1e3:.4f} ms per frame, {rays / dt / 1e6:.1f} Mrays/s")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        main()
