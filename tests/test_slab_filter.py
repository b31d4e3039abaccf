"""The filtered slab test of the WIDE walk (csrc/pt_kernels.hip: slab_q0 / slab_margin / slab_hit), replayed in
float32 on the CPU against the reference's predicate (raytrace.wgsl:118-152).

The kernel decides a box from APPROXIMATE quotients q0 = RN((box - o) * RN(1/d)) and runs the exact
test only when the two ends of the approximate interval lie within 2^-21 (relative) of each other.
The claim (proved in the comment above slab_q0): whenever the margin is positive, the approximate
decision IS the reference's decision.  Checked here on adversarial ray / box pairs -- rays aimed at
box corners and edges (tmin == tmax up to rounding), origins on box faces (tmax == 0), flat boxes
(leaf boxes of axis-aligned triangles: tmin == tmax exactly), boxes far from the origin, direction
components down to the 1e-6 cut-off of the fast path -- and the rate of undecided boxes on
ordinary pairs is bounded (the fallback must stay rare for the filter to pay).

numpy rounds every float32 operation to binary32 and never contracts; the kernel's one fma,
|D| - 2^-21 M, multiplies by a power of two, so the separate multiply + subtract below rounds once
too.  profiles/slab_filter_proof.hip runs the same comparison on the device with the kernel's own
functions (1e11 pairs)."""
import numpy as np

f32 = np.float32
INF = f32(1e20)
BAND = f32(2.0 ** -21)


def exact_hit(o, d, mn, mx):
    """raytrace.wgsl:118-152 for rays / boxes inside the fast path's guards (no parallel axis, no NaN)."""
    with np.errstate(all="ignore"):
        t1 = (mn - o) / d
        t2 = (mx - o) / d
    near, far = np.minimum(t1, t2), np.maximum(t1, t2)
    tmin = np.maximum(np.maximum(np.maximum(-INF, near[:, 0]), near[:, 1]), near[:, 2])
    tmax = np.minimum(np.minimum(np.minimum(INF, far[:, 0]), far[:, 1]), far[:, 2])
    return ~(tmin > tmax) & (tmax >= np.maximum(f32(0), tmin))


def filtered(o, d, mn, mx):
    """slab_q0 + slab_margin + slab_hit: (decided, hit)."""
    y = f32(1) / d                      # rcp_exact == RN(1/d)
    a, b = (mn - o) * y, (mx - o) * y
    near, far = np.minimum(a, b), np.maximum(a, b)
    key = np.maximum(np.maximum(near[:, 0], near[:, 1]), near[:, 2])
    tfar = np.minimum(np.minimum(np.minimum(INF, far[:, 0]), far[:, 1]), far[:, 2])
    D = key - tfar
    M = np.maximum(np.abs(key), np.abs(tfar))
    margin = np.abs(D) - BAND * M
    hit = np.maximum(D, -tfar) <= f32(0)
    return margin > f32(0), hit


def _guards(o, d, mn, mx):
    """the fast path's admission: |d| in [1e-6, 2^20], origin / box coordinates 0 or within [2^-70, 2^60]"""
    ad = np.abs(d)
    ok = ((ad >= f32(1e-6)) & (ad <= f32(1048576.0))).all(1)
    for v in (o, mn, mx):
        av = np.abs(v)
        ok &= ((v == 0) | ((av >= f32(2.0 ** -70)) & (av <= f32(2.0 ** 60)))).all(1)
    return ok


def _check(o, d, mn, mx):
    o, d, mn, mx = (np.ascontiguousarray(v, dtype=f32) for v in (o, d, mn, mx))
    keep = _guards(o, d, mn, mx)
    o, d, mn, mx = o[keep], d[keep], mn[keep], mx[keep]
    want = exact_hit(o, d, mn, mx)
    decided, got = filtered(o, d, mn, mx)
    bad = decided & (got != want)
    assert not bad.any(), f"{int(bad.sum())} decided boxes disagree with the exact test, first: " \
                          f"o={o[bad][0]} d={d[bad][0]} mn={mn[bad][0]} mx={mx[bad][0]}"
    return len(o), int((~decided).sum()), int(want.sum())


def _boxes(rng, n, scale):
    c = rng.uniform(-1, 1, (n, 3)) * scale
    h = np.abs(rng.normal(0, 1, (n, 3))) * scale * rng.choice([1e-3, 1e-2, 0.1, 1.0], (n, 1))
    return (c - h).astype(f32), (c + h).astype(f32)


def _dirs(rng, n):
    d = rng.normal(0, 1, (n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a third of the rays nearly parallel to an axis plane (components down to the fast path's cut-off)
    k = rng.integers(0, 3, n)
    small = rng.random(n) < 0.33
    d[small, k[small]] = rng.choice([-1, 1], small.sum()) * 10.0 ** rng.uniform(-6, -2, small.sum())
    return d.astype(f32)


def test_ordinary_pairs_agree_and_are_rarely_undecided():
    rng = np.random.default_rng(11)
    n = 2_000_000
    total = und = hits = 0
    for scale in (1.0, 5.0, 300.0):
        mn, mx = _boxes(rng, n, scale)
        o = (rng.uniform(-1, 1, (n, 3)) * 2 * scale).astype(f32)
        d = _dirs(rng, n)
        t, u, h = _check(o, d, mn, mx)
        total += t; und += u; hits += h
    assert hits > total // 50
    assert und < total * 1e-4, f"{und} of {total} undecided"      # observed: ~1e-6


def test_rays_through_corners_edges_and_faces():
    """tmin == tmax up to rounding: the ray is aimed (in float64, then rounded) at a corner, an edge point or a
    face point of the box, from outside and from a point ON another face (tmax == 0 / tmin == 0 up to rounding)."""
    rng = np.random.default_rng(12)
    n = 1_500_000
    total = und = 0
    for scale in (1.0, 40.0):
        mn, mx = _boxes(rng, n, scale)
        mn64, mx64 = mn.astype(np.float64), mx.astype(np.float64)
        # target: per axis min, max or a point in between -> corners (3 extremes), edges (2), faces (1)
        pick = rng.integers(0, 3, (n, 3))
        lam = rng.random((n, 3))
        target = np.where(pick == 0, mn64, np.where(pick == 1, mx64, mn64 + lam * (mx64 - mn64)))
        for origin_on_face in (False, True):
            if origin_on_face:
                o64 = mn64 + rng.random((n, 3)) * (mx64 - mn64)
                ax = rng.integers(0, 3, n)
                side = rng.random(n) < 0.5
                o64[np.arange(n), ax] = np.where(side, mn64[np.arange(n), ax], mx64[np.arange(n), ax])
            else:
                o64 = target + rng.normal(0, 1, (n, 3)) * scale * 3
            o = o64.astype(f32)
            d64 = target - o.astype(np.float64)
            nrm = np.linalg.norm(d64, axis=1, keepdims=True)
            good = nrm[:, 0] > 0
            d = np.zeros_like(o)
            d[good] = (d64[good] / nrm[good]).astype(f32)
            # also un-normalised and reversed directions (the box behind the origin: tmax < 0)
            for f in (f32(1), f32(-1), f32(1.7)):
                t, u, _ = _check(o[good], d[good] * f, mn[good], mx[good])
                total += t; und += u
    assert total > 10_000_000
    assert und > 1000          # the construction does reach the band


def test_flat_and_point_boxes_are_undecided_or_right():
    """mn == mx on one, two or three axes (the floor's two triangles; axis-aligned quads): through the flat face the
    approximate interval has zero length -- never decided wrongly."""
    rng = np.random.default_rng(13)
    n = 1_000_000
    mn, mx = _boxes(rng, n, 2.5)
    flat = rng.integers(1, 8, n)
    for k in range(3):
        sel = (flat >> k) & 1 == 1
        mx[sel, k] = mn[sel, k]
    o = (rng.uniform(-1, 1, (n, 3)) * 4).astype(f32)
    o[:, 1] = np.abs(o[:, 1]) + f32(0.5)
    tgt = mn.astype(np.float64) + rng.random((n, 3)) * (mx.astype(np.float64) - mn.astype(np.float64))
    d64 = tgt - o.astype(np.float64)
    d = (d64 / np.linalg.norm(d64, axis=1, keepdims=True)).astype(f32)
    t, u, h = _check(o, d, mn, mx)
    assert h > t // 4 and u > t // 4        # hit through the flat face => zero-length interval => the exact test runs


def test_extreme_magnitudes_inside_the_guards():
    rng = np.random.default_rng(14)
    n = 1_000_000
    e = rng.uniform(-60, 55, (n, 1))
    mn, mx = _boxes(rng, n, 1.0)
    s = (2.0 ** e)
    mn, mx = (mn * s).astype(f32), (mx * s).astype(f32)
    o = (rng.uniform(-1, 1, (n, 3)) * 3 * s).astype(f32)
    d = _dirs(rng, n) * f32(2.0) ** rng.integers(-3, 18, (n, 1)).astype(f32)
    t, u, h = _check(o, d, mn, mx)
    assert t > n // 2
