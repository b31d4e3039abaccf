"""The oracle's pinned transcendental functions against float64 libm (a few ULP), and the
exact-division identity used by the product's slab test (brute force on the CPU)."""
import ctypes

import numpy as np
import pytest


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp[ulp == 0] = 1.4e-45
    return np.abs(got.astype(np.float64) - ref64) / ulp


@pytest.mark.parametrize("fn,name,ref,lo,hi,bound", [
    (0, "sin", np.sin, -10.0, 10.0, 2.0), (1, "cos", np.cos, -10.0, 10.0, 2.0),
    (2, "tan", np.tan, -1.5, 1.5, 4.0), (3, "log", np.log, 1e-30, 1.0, 1.5), (3, "log-big", np.log, 1.0, 1e9, 1.5),
    (4, "exp", np.exp, -87.0, 88.0, 1.5), (6, "asin", np.arcsin, -1.0, 1.0, 3.0),
])
def test_unary_functions_within_a_few_ulp(orc, fn, name, ref, lo, hi, bound):
    rng = np.random.default_rng(fn)
    x = rng.uniform(lo, hi, 400000).astype(np.float32)
    x = x[x != 0] if fn == 3 else x
    e = ulp_err(orc.math_fn(fn, x), ref(x.astype(np.float64)))
    assert e.max() <= bound, (name, e.max(), x[e.argmax()])


def test_atan2_pow_and_specials(orc):
    rng = np.random.default_rng(9)
    y, x = rng.normal(size=400000).astype(np.float32), rng.normal(size=400000).astype(np.float32)
    e = ulp_err(orc.math_fn(5, y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64)))
    assert e.max() <= 4.0
    b = rng.uniform(0, 1, 200000).astype(np.float32)
    g = np.full_like(b, 1 / 2.2)
    e = ulp_err(orc.math_fn(7, b, g), np.power(b.astype(np.float64), np.float64(np.float32(1 / 2.2))))
    assert e.max() <= 8.0
    sp = orc.math_fn(3, np.array([0.0, -1.0, np.inf, 1.0], np.float32))
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and sp[3] == 0.0
    ex = orc.math_fn(4, np.array([-np.inf, -200.0, 0.0, 89.0, np.inf], np.float32))
    assert ex[0] == 0 and ex[1] == 0 and ex[2] == 1 and ex[3] == np.inf and ex[4] == np.inf
    assert orc.math_fn(7, np.array([0.0], np.float32), np.array([0.4545], np.float32))[0] == 0.0     # pow(0, g) = 0
    at = orc.math_fn(5, np.array([0.0, 1.0, -1.0, 0.0], np.float32), np.array([0.0, 0.0, 0.0, -1.0], np.float32))
    assert at[0] == 0 and np.isclose(at[1], np.pi / 2) and np.isclose(at[2], -np.pi / 2) and np.isclose(at[3], np.pi)
    # sin/cos quadrant bookkeeping at exact multiples of pi/2
    q = np.arange(-8, 9, dtype=np.float32) * np.float32(np.pi / 2)
    assert np.allclose(orc.math_fn(0, q), np.sin(q.astype(np.float64)), atol=1e-6)
    assert np.allclose(orc.math_fn(1, q), np.cos(q.astype(np.float64)), atol=1e-6)


def test_fp16_round_trip_is_round_to_nearest_even(orc):
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.normal(size=300000) * 10.0 ** rng.uniform(-9, 6, 300000),
                        [0, -0.0, 65504, 65519.9, 65520, 1e-8, np.inf, -np.inf, 5.96e-8, 2.98e-8, 2.9802322e-8,
                         2.9802326e-8, 6.1e-5, 6.097e-5]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    got = orc.math_fn(8, x)
    assert np.array_equal(got, want)


def test_exact_division_with_a_prepared_reciprocal(orc):
    """q1 = fma(fma(-d,q0,n),y,q0) with y = RN(1/d), q0 = RN(n*y) is the correctly rounded quotient
    over the exponent range the kernel's guards admit, incl. divisors whose significand is all
    ones ('Exact slab test without divisions'; the exhaustive significand-pair check
    is profiles/div_proof.hip, run on the device)."""
    lib = orc.lib()
    lib.orc_check_div_pre.restype = ctypes.c_uint64
    lib.orc_check_div_pre.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    assert lib.orc_check_div_pre(60_000_000, 12345) == 0
