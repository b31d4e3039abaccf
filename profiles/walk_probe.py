#!/usr/bin/env python3
"""Design experiment: how fast is the BVH walk ALONE, and what would more resident waves buy?

Builds a realistic ray mix for the 1080p demo frame (all primary rays + three generations of
diffuse bounce rays started at the hit points the device reports), then times the walk-only
persistent kernel (mi3pt_debug_walk_probe: 56 VGPRs, 4 KB LDS) at 4, 5, 6 and 8 waves per SIMD.
The full kernel (120 VGPRs) can only hold 4.  usage: python profiles/walk_probe.py [WxH]  (needs a GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import ptcommon as pc  # noqa: E402
from mi3pt_host import capi, scenes  # noqa: E402

w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
workload = os.environ.get("WORKLOAD", "demo")
sc = scenes.dragon_class_scene() if workload == "dragon" else scenes.demo_scene()
sc.build_bvh()
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, scenes.synthetic_env())
rng = np.random.default_rng(5)

# primary rays: pinhole through pixel centres (the shader's camera quirks do not matter here)
cam = np.array(sc.camera["position"], np.float64)
fwd = sc.camera_direction()
right = np.cross(fwd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
up = np.cross(right, fwd)
t = np.tan(np.radians(sc.camera["fov"]) / 2)
ys, xs = np.mgrid[0:h, 0:w]
# tile-major order, like the kernel hands pixels out
order = np.lexsort((xs.ravel() % 8, ys.ravel() % 8, xs.ravel() // 8, ys.ravel() // 8))
u = ((xs.ravel()[order] + 0.5) / w * 2 - 1) * t * (w / h)
v = ((ys.ravel()[order] + 0.5) / h * 2 - 1) * t
d = fwd[None] * (w / h) + right[None] * u[:, None] + up[None] * v[:, None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([np.broadcast_to(cam, d.shape), d], axis=1).astype(np.float32)
generations = [rays]
for g in range(3):
    ms, hits = ctx.walk_probe(generations[-1], 4 if workload == "demo" else 105, repeats=1, want_hits=True)      # (the legacy probe's 16-entry stack does not hold the large tree)
    hit = hits[:, 3].view(np.int32) >= 0
    src = generations[-1][hit]
    pos = src[:, :3] + src[:, 3:] * hits[hit, 0:1]
    # diffuse bounce: uniform direction on the sphere around a random "normal" side -- incoherent like the real thing
    nd = rng.normal(size=(len(pos), 3)).astype(np.float32)
    nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    nd[:, 1] = np.abs(nd[:, 1])                 # most surfaces of the demo scene face up
    generations.append(np.concatenate([pos, nd], axis=1).astype(np.float32))
mix = np.concatenate(generations)
print(f"{workload} {w}x{h}: ray mix {len(mix)} rays ({', '.join(str(len(g)) for g in generations)} per generation)")
base = None
for waves in ((4, 5, 6, 8) if workload == "demo" else ()):
    ms, _ = ctx.walk_probe(mix, waves, repeats=3, passes=16)
    base = base or ms
    print(f"  {waves} waves per SIMD: {ms:.3f} ms for 16 passes  {16 * len(mix) / ms / 1e3:.0f} Mrays/s walk-only  ({base / ms:.2f}x the 4-wave rate)")
# the SHIPPED walk alone (round 5: compressed wide packets, distance culling, paired triangle steps -- k_walk_probe_cw), every idle lane
# refilled at once: what a wave that does nothing but walk sustains, and whether its hits are the reference walk's
if workload == "demo":
    _, ref_hits = ctx.walk_probe(mix, 4, repeats=1, want_hits=True)
else:                                   # the per-ray reference walk (t, u, v of the closest hit: columns 1 .. of mi3pt_debug_intersect are position-based; compare t only)
    ref_hits = None
for refill_min in (1, 8, 16, 24, 32):          # idle lanes that trigger a refill (each refill is a memory round trip for the wave, like a service step)
    ctx.set_option(capi.OPT_TOP_PACKETS, refill_min)
    for waves in (5, 6, 7):
        ms, hits = ctx.walk_probe(mix, 100 + waves, repeats=3, passes=16, want_hits=True)
        note = capi.last_error()
        same = np.array_equal(hits.view(np.uint32), ref_hits.view(np.uint32)) if ref_hits is not None else "n/a"
        print(f"  compressed-wide walk, refill at {refill_min:2d} idle lanes, {waves} waves per SIMD: {ms:.3f} ms for 16 passes  {16 * len(mix) / ms / 1e3:.0f} Mrays/s walk-only   "
              f"[{note}]   hits identical to the reference walk's: {same}"
              + ("" if same in (True, "n/a") else f" ({int((hits.view(np.uint32) != ref_hits.view(np.uint32)).any(axis=1).sum())} rays differ)"), flush=True)
# reference point: the coherent primary rays alone
for waves in ((4, 8) if workload == "demo" else ()):
    ms, _ = ctx.walk_probe(generations[0], waves, repeats=3, passes=16)
    print(f"  primary rays only, {waves} waves per SIMD: {ms:.3f} ms for 16 passes  {16 * len(generations[0]) / ms / 1e3:.0f} Mrays/s")
# cost of the mix in the reference's own unit: box tests per ray (the full kernel's frames average 10.4 on this scene)
for name, g in (("primary", generations[0]), ("bounce 1", generations[1]), ("bounce 2", generations[2]), ("bounce 3", generations[3])):
    sample = g[rng.choice(len(g), size=min(len(g), 200000), replace=False)]
    res = ctx.debug_intersect(sample)
    print(f"  {name}: {res[:, 9].mean():.1f} box tests and {res[:, 10].mean():.2f} triangle tests per ray, {100 * res[:, 0].mean():.0f} % hit")
