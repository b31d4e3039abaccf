#!/usr/bin/env python3
"""What bounds the forest's 235 box tests per ray (round-5 verdict, next #7): statistics of the REFERENCE tree of BASELINE.json
config 5's scene (the tree every walk of this library is derived from) and of where a sample of rays spends its box tests.
CPU only (the host-side builder + a plain Python walk of the reference's rayBVHIntersect, raytrace.wgsl:154-203, without
culling); prints a log.   usage: python profiles/forest_tree_stats.py [rays, default 1500] [instances, default 9000]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
import numpy as np
from mi3pt_host import scenes

nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
inst = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
t0 = time.time()
sc = scenes.forest_scene(instances=inst)
sc.build_bvh()
nodes = sc.nodes
n = len(nodes)
mn = np.array(nodes["min"], np.float64); mx = np.array(nodes["max"], np.float64)
left = np.array(nodes["left"], np.int64); right = np.array(nodes["right"], np.int64); leaf = np.array(nodes["isLeaf"], np.int64) == 1
tri = np.array(nodes["triangleIndex"], np.int64)
print(f"# forest: {len(sc.triangles)} triangles, {n} nodes ({time.time() - t0:.0f} s to generate and build)")
ext = mx - mn
area = 2.0 * (ext[:, 0] * ext[:, 1] + ext[:, 0] * ext[:, 2] + ext[:, 1] * ext[:, 2])
# depth of every node (BFS numbering: children after their parent)
depth = np.zeros(n, np.int64)
internal = np.nonzero(~leaf)[0]
for i in internal:
    depth[left[i]] = depth[i] + 1; depth[right[i]] = depth[i] + 1
ld = depth[leaf]
print(f"# leaf depth: min {ld.min()}  median {int(np.median(ld))}  mean {ld.mean():.1f}  p95 {int(np.percentile(ld, 95))}  max {ld.max()}   (a balanced tree of {leaf.sum()} leaves: {math.log2(leaf.sum()):.1f})")
hist = np.bincount(ld)
print("# leaf-depth histogram (depth: share of leaves): " + "  ".join(f"{d}: {100.0 * c / leaf.sum():.1f}%" for d, c in enumerate(hist) if c * 200 > leaf.sum()))
sah_int = area[~leaf].sum() / area[0]; sah_leaf = area[leaf].sum() / area[0]
print(f"# SAH (surface-area sums over the root's area): internal nodes {sah_int:.1f}  leaves {sah_leaf:.1f}   = expected nodes / leaves a random line through the root box enters")
# overlap: how many LEAF boxes and how many tree-sized internal boxes (diagonal 1 .. 4 units: a crown) contain a random point of the canopy layer
rng = np.random.default_rng(3)
diag = np.sqrt((ext ** 2).sum(1))
crown = (~leaf) & (diag > 1.0) & (diag < 4.0)
pts = np.stack([rng.uniform(-18, 18, 400), rng.uniform(0.6, 1.6, 400), rng.uniform(-18, 18, 400)], 1)
def containing(sel):
    a, b = mn[sel], mx[sel]
    return np.array([((a <= p) & (p <= b)).all(1).sum() for p in pts])
cc = containing(crown)
print(f"# overlap: a random point of the canopy layer (y 0.6 .. 1.6) lies inside {cc.mean():.1f} internal boxes of crown size (diagonal 1 .. 4; max {cc.max()}) -- "
      f"the tree instances' crowns interpenetrate ({inst} trees of radius ~0.5 .. 1 on a {40}x{40} square = {inst / 1600.0:.1f} per unit area)")

# ---- a sample of rays: camera rays of the bench view + one diffuse bounce each, walked like the reference (no culling), per-depth box tests
cam = np.array(sc.camera["position"], np.float64); tgt = np.array(sc.camera["target"], np.float64)
fwd = tgt - cam; fwd /= np.linalg.norm(fwd)
rightv = np.cross(fwd, [0, 1, 0]); rightv /= np.linalg.norm(rightv); up = np.cross(rightv, fwd)
fov = math.radians(sc.camera["fov"]); aspect = 1920 / 1080
pos = np.array(sc.positions, np.float64).reshape(-1, 3, 3) if hasattr(sc, "positions") else None

def slab(o, inv, i):
    t1 = (mn[i] - o) * inv; t2 = (mx[i] - o) * inv
    tmin = np.minimum(t1, t2).max(); tmax = np.maximum(t1, t2).min()
    return tmax >= max(0.0, tmin)

def walk(o, d):
    """box tests by depth, triangle candidates, closest hit (t, triangle)"""
    inv = 1.0 / np.where(np.abs(d) < 1e-9, 1e-9, d)
    by_depth = {}
    best_t, best_tri = 1e30, -1
    if not slab(o, inv, 0):
        return {0: 1}, 0, best_t, best_tri
    by_depth[0] = 1
    stack = [0]; ntri = 0
    while stack:
        i = stack.pop()
        if leaf[i]:
            ntri += 1
            a, b, c = pos[tri[i]]
            e1, e2 = b - a, c - a
            h = np.cross(d, e2); det = e1.dot(h)
            if abs(det) > 1e-6:
                f = 1.0 / det; s = o - a; u = f * s.dot(h)
                if 0 <= u <= 1:
                    q = np.cross(s, e1); v = f * d.dot(q)
                    if v >= 0 and u + v <= 1:
                        t = f * e2.dot(q)
                        if 1e-6 < t < best_t: best_t, best_tri = t, int(tri[i])
            continue
        dd = int(depth[i]) + 1
        for c in (left[i], right[i]):
            by_depth[dd] = by_depth.get(dd, 0) + 1
            if slab(o, inv, c): stack.append(c)
    return by_depth, ntri, best_t, best_tri

tot = {"camera": {}, "bounce": {}}; cnt = {"camera": 0, "bounce": 0}; tris = {"camera": 0, "bounce": 0}; hits = 0
t1 = time.time()
for k in range(nrays):
    u, v = rng.uniform(0, 1), rng.uniform(0, 1)
    x = (2 * u - 1) * math.tan(fov / 2) * aspect; y = (1 - 2 * v) * math.tan(fov / 2)
    d = fwd + x * rightv + y * up; d /= np.linalg.norm(d)
    bd, nt_, t, ti = walk(cam, d)
    for dep, c in bd.items(): tot["camera"][dep] = tot["camera"].get(dep, 0) + c
    cnt["camera"] += 1; tris["camera"] += nt_
    if ti >= 0:
        hits += 1
        a, b, c = pos[ti]; nrm = np.cross(b - a, c - a); nrm /= max(np.linalg.norm(nrm), 1e-30)
        if nrm.dot(d) > 0: nrm = -nrm
        r = rng.normal(size=3); r /= np.linalg.norm(r); nd = nrm + r; nd /= max(np.linalg.norm(nd), 1e-30)
        bd, nt_, _, _ = walk(cam + d * t, nd)
        for dep, c in bd.items(): tot["bounce"][dep] = tot["bounce"].get(dep, 0) + c
        cnt["bounce"] += 1; tris["bounce"] += nt_
print(f"# {nrays} camera rays of the bench view ({hits} hit) + one diffuse bounce each, the reference's walk without culling ({time.time() - t1:.0f} s):")
for kind in ("camera", "bounce"):
    if not cnt[kind]: continue
    total = sum(tot[kind].values())
    print(f"#   {kind:6s}: {total / cnt[kind]:.0f} box tests + {tris[kind] / cnt[kind]:.1f} triangle tests per ray;  box tests by node depth: " +
          "  ".join(f"{lo}-{hi}: {100.0 * sum(c for d_, c in tot[kind].items() if lo <= d_ <= hi) / total:.0f}%" for lo, hi in ((0, 7), (8, 13), (14, 19), (20, 25), (26, 31), (32, 99))))
