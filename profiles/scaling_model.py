#!/usr/bin/env python3
"""Predicts the strong-scaling curve of bench.py from ONE GPU.  The ranks of a tile split share
nothing but the final gather (scene replicated, no per-frame communication), so the time of an
N-GPU job is the slowest rank's time plus the gather.  Each rank's share is rendered alone on
this GPU (bench.py --tile R/N) with the driver's arguments (a step = 16 frames); the gather is priced from the
payload (33 MB / N per rank, point-to-point to rank 0 over distinct xGMI links, ~45 GB/s
effective per link + ~60 us of launch / rendezvous, profiles/r01_g_gather_bench.log).
usage: python profiles/scaling_model.py [--steps 20 --warmup 5 --workload dragon]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
only = None
if "--only" in sys.argv:          # --only 8: that split alone (with MI3PT_BENCH_BLOCK_ROWS=16 / 32: the block size of the deal)
    i = sys.argv.index("--only"); only = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
extra = sys.argv[1:] or ["--steps", "20", "--warmup", "5"]


def run(tile):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-pmc", "--no-cpu-baseline", "--no-also", *extra]
    if tile:
        cmd += ["--tile", tile]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600).stdout
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


base = run(None)
print(f"N=1: {base['value']:.0f} Mrays/s, {base['ms_per_step'] * base['steps']:.3f} ms for {base['steps']} steps", flush=True)
for n in ((only,) if only else (2, 4, 8)):
    ranks = [run(f"{r}/{n}") for r in range(n)]          # every rank (round 4 rendered three of the eight)
    t = [j["ms_per_step"] * j["steps"] for j in ranks]
    rays = sum(j["config"]["rays_per_step"] for j in ranks) * (n / len(ranks)) * base["steps"]
    gather_ms = 0.06 + (1920 * 1080 * 16 / n) * (n - 1) / (min(n - 1, 7) * 45e9) * 1e3
    total = max(t) + gather_ms
    val = rays / total / 1e3
    print(f"N={n}: rank times {', '.join(f'{x:.3f}' for x in t)} ms (+ gather {gather_ms:.3f} ms) -> {val:.0f} Mrays/s, "
          f"speed-up {val / base['value']:.2f}x, efficiency {val / base['value'] / n:.0%}", flush=True)
