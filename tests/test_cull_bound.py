"""The error bound behind the distance-culling walk (kernel variant 9), checked
numerically on the CPU: the reference's Moller-Trumbore arithmetic (raytrace.wgsl:78-116) is
replayed in float32, operation by operation as the kernel and the oracle perform it, on
adversarial ray / triangle pairs -- grazing incidence down to the 1e-6 determinant cut-off,
slivers, coordinates far from the origin, un-normalised directions.  For every pair the code
ACCEPTS with distance t, the point o + t d must lie within
    delta(E, L, t) = W(E) (u / EPSILON) (t |d|^2 + 1.65 L |d|),   u = 2^-24,
of the triangle (E = |e1||e2|, L = |e1| + |e2|, W as in csrc/pt_context.hip prepare_cull), and the
kernel's skip predicate, evaluated with the triangle's own bounding box, must not fire for
best.t >= t.  A proof bounds the worst case; this test shows how much room the constants leave
(observed: about 1/8 of the bound) and guards the formulas against transcription errors."""
import numpy as np

f32 = np.float32
U = 2.0 ** -24
EPS = float(f32(1e-6))


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                     a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1)


def _dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def moller_trumbore_f32(o, d, a, b, c):
    """raytrace.wgsl:78-116 in float32 (numpy rounds every operation to binary32)."""
    e1, e2 = b - a, c - a
    h = _cross(d, e2)
    det = _dot(e1, h)
    ok = ~((det > -f32(1e-6)) & (det < f32(1e-6)))
    with np.errstate(all="ignore"):
        f = f32(1) / det
        s = o - a
        u = f * _dot(s, h)
        ok &= ~((u < 0) | (u > 1))
        q = _cross(s, e1)
        v = f * _dot(d, q)
        ok &= ~((v < 0) | (u + v > f32(1)))
        t = f * _dot(e2, q)
    ok &= t > f32(1e-6)
    return ok, t, det


def weight(E):
    """W(E) = E c1(E) of prepare_cull (csrc/pt_context.hip); nan where the triangle is outside the analysis."""
    kappa = E * 2.0 / EPS
    A = 5.85 * U * kappa
    b = (1.0 + A) / (1.0 - A) + 1.0
    den = 1.0 - (11.7 * b + 1.02) * U * kappa
    w = E * (11.7 * b + 3.04) / den
    return np.where((A < 0.25) & (den > 0.5), w, np.nan)


def _point_triangle_distance(p, a, b, c):
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(1), (ac * ap).sum(1)
    bp = p - b
    d3, d4 = (ab * bp).sum(1), (ac * bp).sum(1)
    cp = p - c
    d5, d6 = (ab * cp).sum(1), (ac * cp).sum(1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    out = np.empty_like(p)
    done = np.zeros(len(p), bool)

    def put(mask, val):
        nonlocal done
        m = mask & ~done
        out[m] = val[m]
        done |= m

    with np.errstate(all="ignore"):
        put((d1 <= 0) & (d2 <= 0), a)
        put((d3 >= 0) & (d4 <= d3), b)
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + ab * (d1 / (d1 - d3))[:, None])
        put((d6 >= 0) & (d5 <= d6), c)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + ac * (d2 / (d2 - d6))[:, None])
        put((va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0), b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None])
        den = 1.0 / (va + vb + vc)
        put(np.ones(len(p), bool), a + ab * (vb * den)[:, None] + ac * (vc * den)[:, None])
    return np.sqrt(((p - out) ** 2).sum(1))


def _adversarial_batch(rng, n):
    size = 10 ** rng.uniform(-4, 0.2, n)
    aspect = 10 ** rng.uniform(-3, 0, n)
    centre = rng.normal(size=(n, 3)) * (10 ** rng.uniform(-1, 2.5, n))[:, None]
    u1 = rng.normal(size=(n, 3)); u1 /= np.linalg.norm(u1, axis=1)[:, None]
    u2 = rng.normal(size=(n, 3)); u2 -= (u2 * u1).sum(1)[:, None] * u1; u2 /= np.linalg.norm(u2, axis=1)[:, None]
    ang = rng.uniform(0.05, 3.09, n)
    a = centre.astype(f32)
    b = (centre + u1 * size[:, None]).astype(f32)
    c = (centre + (np.cos(ang)[:, None] * u1 + np.sin(ang)[:, None] * u2) * (size * aspect)[:, None]).astype(f32)
    normal = np.cross(u1, u2)
    r1, r2 = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    m = r1 + r2 > 1
    r1[m], r2[m] = 1 - r1[m], 1 - r2[m]
    p = a.astype(np.float64) + (b - a).astype(np.float64) * r1[:, None] + (c - a).astype(np.float64) * r2[:, None]
    p += (u1 * rng.normal(size=(n, 1)) + u2 * rng.normal(size=(n, 1))) * (size * 10 ** rng.uniform(-8, -1, n))[:, None]
    w = rng.uniform(0, 2 * np.pi, n)
    inplane = np.cos(w)[:, None] * u1 + np.sin(w)[:, None] * u2
    tilt = 10 ** rng.uniform(-7.5, 0.3, n) * rng.choice([-1, 1], n)       # down to grazing at the determinant cut-off
    d = inplane + tilt[:, None] * normal
    d /= np.linalg.norm(d, axis=1)[:, None]
    d *= rng.uniform(0.3, 1.0, n)[:, None]                                 # mix() leaves directions un-normalised
    t0 = 10 ** rng.uniform(-3, 3, n)
    return (p - d * t0[:, None]).astype(f32), d.astype(f32), a, b, c


def test_accepted_hits_stay_within_delta_and_are_never_skipped():
    rng = np.random.default_rng(20251004)
    worst, accepted = 0.0, 0
    for _ in range(4):
        o, d, a, b, c = _adversarial_batch(rng, 400_000)
        ok, t, det = moller_trumbore_f32(o, d, a, b, c)
        o, d, a, b, c, t = (x[ok] for x in (o, d, a, b, c, t))
        o6, d6, a6, b6, c6, t6 = (x.astype(np.float64) for x in (o, d, a, b, c, t))
        e1, e2 = np.linalg.norm(b6 - a6, axis=1), np.linalg.norm(c6 - a6, axis=1)
        E, L = e1 * e2, e1 + e2
        W = weight(E)
        sel = ~np.isnan(W)
        dd = (d6 * d6).sum(1)
        delta = W * (U / EPS) * (t6 * dd + 1.65 * L * np.sqrt(dd))
        dist = _point_triangle_distance(o6 + t6[:, None] * d6, a6, b6, c6)
        assert (dist[sel] <= delta[sel]).all()
        worst = max(worst, float((dist[sel] / delta[sel]).max()))
        accepted += int(sel.sum())

        # the kernel's skip predicate (pt_kernels.hip, CULL node step) with the triangle's own box and best.t = t
        mn, mx = np.minimum(np.minimum(a, b), c), np.maximum(np.maximum(a, b), c)
        with np.errstate(all="ignore"):
            inv = f32(1) / d
            tnear = np.minimum((mn - o) / d, (mx - o) / d)
        parallel = (np.abs(d) < f32(1e-6)).any(1)
        lmax = f32(np.nextafter(f32(L[sel].max() * (1 + 1e-6)), f32(np.inf)))
        scene_ka = f32(np.nextafter(f32(U / EPS * 1.001), f32(np.inf)))
        scene_kb = f32(np.nextafter(f32(1.65 * float(lmax) * U / EPS * 1.001), f32(np.inf)))
        ddf = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) * f32(1.0000005)
        ka, kb = scene_ka * ddf, scene_kb * np.maximum(f32(1), ddf)
        w16 = (np.where(sel, W, 0.0) * (1 + 1e-6)).astype(f32)
        w16 = ((w16.view(np.uint32) + np.uint32(0xffff)) & np.uint32(0xffff0000)).view(f32)      # the packet's 16 bits, rounded up
        dl = w16 * (ka * t + kb)
        with np.errstate(all="ignore"):
            tc = np.max(tnear - dl[:, None] * np.abs(inv), axis=1)
        skipped = (tc > t * f32(1.00000095367431640625)) & sel & ~parallel
        assert not skipped.any()
    assert accepted > 200_000
    assert worst < 0.5          # the proof's constants leave room (observed ~0.12)


def test_tiny_triangles_are_invisible_to_the_reference():
    """|det| <= |e1||e2||d| (1 + 7u): a triangle with E |d| below the 1e-6 cut-off is rejected by
    raytrace.wgsl:88-90 whatever the ray -- the analysis may give it weight W(E) ~ 0."""
    rng = np.random.default_rng(7)
    n = 200_000
    o, d, a, b, c = _adversarial_batch(rng, n)
    scale = (10 ** rng.uniform(-6, -3.5, n)).astype(f32)[:, None]
    b, c = a + (b - a) * scale, a + (c - a) * scale
    E = np.linalg.norm((b - a).astype(np.float64), axis=1) * np.linalg.norm((c - a).astype(np.float64), axis=1)
    dn = np.linalg.norm(d.astype(np.float64), axis=1)
    ok, _, _ = moller_trumbore_f32(o, d, a, b, c)
    assert not ok[E * dn < 0.99e-6].any()
