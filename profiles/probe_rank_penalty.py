#!/usr/bin/env python3
"""Why does a rank of an 8-way split cost more per pixel than the whole image?  (round-3 verdict: 32.95 us per frame in
steady state where an eighth of the whole image is 28.8.)

Legs, each `bench.py --steps S --warmup W` on this one GPU (deep jobs: ramp and drain are amortised):
  whole            the whole 1080p image, with the four counter passes
  3/8 rows=8       rank 3 of the 8-way split, 8-row blocks round robin (the shipped split), with counter passes
  R/8 rows=135     the EQUAL CONTIGUOUS split (block_rows = ceil(1080 / 8): rank R owns rows [135 R, 135 R + 135)), R = 0..7
  3/8 rows=B       the interleave granularity: B = 1, 2, 4, 16, 32, 64
Printed per leg: Mrays/s, us per frame, rays per frame, and per RAY: HBM-side bytes, L2 requests, L2 hit rate, VALU
instructions, wave cycles waiting for memory -- the quantities the hypotheses differ in.
usage: python profiles/probe_rank_penalty.py [pmc|nopmc]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARM = 64, 32
with_pmc = (sys.argv[1:] or ["pmc"])[0] == "pmc"


def run(tile, rows, pmc):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-also", "--no-forest", "--steps", str(STEPS), "--warmup", str(WARM)]
    if not pmc:
        cmd.append("--no-pmc")
    if tile:
        cmd += ["--tile", tile]
    env = dict(os.environ, MI3PT_BENCH_BLOCK_ROWS=str(rows))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print(f"  {tile} rows={rows}: no line: {out.stderr[-300:]}", flush=True)
        return None
    return json.loads(lines[-1])


def show(name, j):
    if j is None:
        return
    frames = j["steps"] * 16
    rays_frame = j["config"]["rays_per_step"] / 16.0
    us_frame = j["ms_per_step"] / 16.0 * 1e3
    s = f"{name:16s} {j['value']:8.0f} Mrays/s  {us_frame:7.2f} us/frame  {rays_frame / 1e3:8.1f} k rays/frame  {us_frame / rays_frame * 1e6:6.2f} ps/ray"
    r = j.get("roofline") or {}
    c = r.get("pmc_counters") or {}
    if c:
        rays_launch = rays_frame * r["frames_per_launch"]
        per = lambda k: c.get(k, float("nan")) / rays_launch
        s += (f" | per ray: HBM-side {(2 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024 / rays_launch:6.1f} B (write {c.get('WRITE_SIZE', 0) * 1024 / rays_launch:5.1f})"
              f"  L2 req {per('TCC_HIT_sum') + per('TCC_MISS_sum'):6.2f} hit {c.get('TCC_HIT_sum', 0) / max(c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0), 1):.3f}"
              f"  VALU {per('SQ_INSTS_VALU'):6.1f}  lane util {r.get('lane_utilisation')}  wave cycles {per('SQ_WAVE_CYCLES'):7.0f} (wait mem {per('SQ_WAIT_ANY'):7.0f}, issue stall {per('SQ_WAIT_INST_ANY'):6.0f}, issuing {per('SQ_ACTIVE_INST_ANY'):6.0f})"
              f"  kernel ms excl {r.get('kernel_ms_exclusive')} x {r.get('launches_timed')} launches of {r.get('frames_per_launch')} frames")
    print(s, flush=True)


show("whole", run(None, 8, with_pmc))
show("3/8 rows=8", run("3/8", 8, with_pmc))
for r in range(8):
    show(f"{r}/8 rows=135", run(f"{r}/8", 135, with_pmc and r == 3))
for b in (1, 4, 32):
    show(f"3/8 rows={b}", run("3/8", b, False))
for r in (0, 7):
    show(f"{r}/8 rows=8", run(f"{r}/8", 8, False))
