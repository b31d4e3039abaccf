// pt_devmath.h -- fp32 math used by the gfx950 kernels.
//
// The reference's WGSL leaves the ULP behaviour of sin/cos/tan/log/exp/atan2/asin/pow
// and FMA contraction implementation-defined (raytrace.wgsl:218, 262-264, 284-286,
// 294-304; fullscreen.wgsl:73-79, 102).  DESIGN.md "Pinned arithmetic" fixes one
// interpretation: IEEE binary32, round-to-nearest-even, denormals kept, correctly
// rounded + - * / sqrt, no contraction except where fmaf is written, and Cephes-style
// single-precision range reduction + minimax polynomials in Horner/fmaf form.
// These are the device implementations of that specification.  The translation unit
// MUST be built with -ffp-contract=off and without fast-math.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

namespace ptm {

PT_DEV float pow2i(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }

// ---------------------------------------------------------------------------------
// Correctly rounded division, reciprocal and square root in fewer instructions than the
// compiler's IEEE expansions (11 / 11 / 14).  Each identity was checked by exhaustion on
// gfx950 (profiles/div_proof.hip; logs under profiles/r01_g_*_proof_exhaustive.log) and is
// only applied where no intermediate can leave the normal range; outside, the plain
// operation runs.  Results are bit-identical to '/', 1.0f / x and sqrtf.
// ---------------------------------------------------------------------------------

// RN(n/d) given y = RN(1/d): all 2^23 x 2^23 significand pairs checked; the caller guarantees
// that n is 0 or large enough for the residual to be exact (|n| >= 2^-103) and that n/d stays
// within the normal range.
PT_DEV float div_pre(float n, float d, float y)
{
    const float q0 = n * y;
    return fmaf(fmaf(-d, q0, n), y, q0);
}

// RN(1/d): one Newton step on v_rcp_f32; all 2^32 inputs checked, exact for |d| in [2^-64, 2^64].
PT_DEV float rcp_exact(float d)
{
    const float a = fabsf(d);
    if (!(a >= 5.421010862427522e-20f && a <= 1.8446744073709552e19f)) return 1.0f / d;
    const float y0 = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, y0, 1.0f), y0, y0);
}

// RN(sqrt x): y = v_rsq_f32(x), s = x*y, s + (x - s*s) * (y/2); all non-negative inputs
// checked, exact for x in [2^-64, 2^64] (v_sqrt_f32 alone is off for 3.3e8 inputs).
PT_DEV float sqrt_exact(float x)
{
    if (!(x >= 5.421010862427522e-20f && x <= 1.8446744073709552e19f)) return sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    return fmaf(fmaf(-s, s, x), 0.5f * y, s);
}

// z * 2^n, single rounding, n in [-150, 254]
PT_DEV float ldexp1(float z, int n)
{
    if (n > 127) {
        z = z * pow2i(127);
        n -= 127;
        if (n > 127) n = 127;
        return z * pow2i(n);
    }
    if (n < -126) {
        n += 24;
        if (n < -126) n = -126;
        return (z * pow2i(n)) * pow2i(-24);
    }
    return z * pow2i(n);
}

// sin and cos together: Cody-Waite reduction by pi/2 in three parts, then the
// degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4].
PT_DEV void sincos(float x, float &s_out, float &c_out)
{
    const float q = rintf(x * 0.636619772367581343f);
    float r = fmaf(-q, 1.5703125f, x);
    r = fmaf(-q, 4.837512969970703125e-4f, r);
    r = fmaf(-q, 7.54978995489188e-8f, r);
    const int n = (int)q;
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float s = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float c = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    const bool swap = (n & 1) != 0;
    float rs = swap ? c : s;
    float rc = swap ? s : c;
    // quadrant signs: n&3 = 0:(s,c) 1:(c,-s) 2:(-s,-c) 3:(-c,s)
    if (n & 2) rs = -rs;
    if (((n + 1) & 2) != 0) rc = -rc;
    s_out = rs;
    c_out = rc;
}
PT_DEV float sin1(float x) { float s, c; sincos(x, s, c); return s; }
PT_DEV float cos1(float x) { float s, c; sincos(x, s, c); return c; }
PT_DEV float tan1(float x) { float s, c; sincos(x, s, c); return s / c; }

PT_DEV float log1(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return __uint_as_float(0x7fc00000u);
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int e = 0;
    uint32_t u = __float_as_uint(x);
    if ((u >> 23) == 0) {
        x = x * 33554432.0f;
        u = __float_as_uint(x);
        e = -25;
    }
    e += (int)(u >> 23) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    const float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

// log1 for x = 0 or x in [2^-32, 1] -- everything rand() returns (raytrace.wgsl:253-259: an integer below 2^32, rounded to
// binary32, times 2^-32): never a NaN, negative, infinite or subnormal, so log1's tests for those and its subnormal scaling
// drop out; the remaining operations are log1's own, in its order.
PT_DEV float log1_unit(float x)
{
    const uint32_t u = __float_as_uint(x);
    int e = (int)(u >> 23) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    const float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return x == 0.0f ? -__builtin_inff() : r;
}

PT_DEV float exp1(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return __builtin_inff();
    if (x < -103.972077083991796f) return 0.0f;
    const float k = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    float r = fmaf(-k, 0.693359375f, x);
    r = fmaf(-k, -2.12194440e-4f, r);
    const float z = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, z, r) + 1.0f;
    return ldexp1(y, (int)k);
}

// exp1 for x <= 0 (or NaN): the same value from fewer instructions.  No overflow side; the scaling by 2^k (k in
// [-150, 0]) is one v_ldexp_f32, which rounds a subnormal result once like ldexp1's two-step product does (f32 subnormals
// are kept in this library's kernels); and exp1's cut to 0 below -103.97 comes by itself -- there k <= -150 and the
// polynomial's value is <= 1, so the scaled value is at most half the smallest subnormal and rounds to 0 (a tie goes to
// the even 0) -- once -inf is kept out of the reduction (the argument is held at -200: same k range, same 0).
// All 2^31 + 1 non-positive inputs and every NaN compared with exp1 on the device: profiles/exp_nonpos_proof.hip.
PT_DEV float exp1_nonpos(float x)
{
    const float xc = __builtin_fmaxf(x, -200.0f);
    const float k = floorf(fmaf(1.44269504088896341f, xc, 0.5f));
    float r = fmaf(-k, 0.693359375f, xc);
    r = fmaf(-k, -2.12194440e-4f, r);
    const float z = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, z, r) + 1.0f;
    const float e = __builtin_ldexpf(y, (int)k);
    return x != x ? x : e;
}

PT_DEV float pow1(float x, float y) { return exp1(y * log1(x)); }

PT_DEV float atan1(float x)
{
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) {
        y = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    const float z = x * x;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return sign * y;
}

PT_DEV float atan2_1(float y, float x)
{
    const float PI_F = 3.14159265358979323846f;
    const float PIO2_F = 1.5707963267948966f;
    if (x != x || y != y) return __uint_as_float(0x7fc00000u);
    if (x == 0.0f) {
        if (y > 0.0f) return PIO2_F;
        if (y < 0.0f) return -PIO2_F;
        return 0.0f;
    }
    const float a = atan1(y / x);
    if (x > 0.0f) return a;
    if (y >= 0.0f) return a + PI_F;
    return a - PI_F;
}

PT_DEV float asin1(float x)
{
    float sign = 1.0f;
    float a = x;
    if (x < 0.0f) { sign = -1.0f; a = -x; }
    if (a != a) return a;
    if (a > 1.0f) return __uint_as_float(0x7fc00000u);
    if (a < 1.0e-4f) return x;
    bool flag = false;
    float z, t;
    if (a > 0.5f) {
        z = 0.5f * (1.0f - a);
        t = sqrt_exact(z);
        flag = true;
    } else {
        t = a;
        z = t * t;
    }
    float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    float r = fmaf(p * z, t, t);
    if (flag) {
        r = r + r;
        r = 1.5707963267948966f - r;
    }
    return sign * r;
}

// rgba16float storage: round-to-nearest-even through binary16 (v_cvt_f16_f32 /
// v_cvt_f32_f16 are IEEE conversions, denormal halves kept).
PT_DEV float round_f16(float f) { return __half2float(__float2half_rn(f)); }

}  // namespace ptm
