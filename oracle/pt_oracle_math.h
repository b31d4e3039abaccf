/*
 * pt_oracle_math.h -- TEST INFRASTRUCTURE (oracle). Not product code.
 *
 * Pinned fp32 interpretation of the WGSL builtins the reference's shaders call
 * (src/passes/shaders/raytrace.wgsl, accumulate.wgsl, fullscreen.wgsl).  WGSL leaves
 * the ULP behaviour of sin/cos/tan/log/exp/atan2/asin/pow and FMA contraction
 * implementation-defined, so "the reference's output" only exists up to those
 * choices.  This header fixes ONE choice (see DESIGN.md "Pinned arithmetic"):
 *
 *   - every operation is IEEE-754 binary32, round-to-nearest-even, denormals kept;
 *   - `+ - * / sqrt` are correctly rounded, never contracted (build with
 *     -ffp-contract=off); fused multiply-add happens ONLY where fmaf() is written;
 *   - transcendental functions are the classic Cephes single-precision
 *     range-reduction + minimax polynomials (Moshier, netlib cephes/single),
 *     evaluated in Horner form with fmaf(), restated here from the published
 *     algorithm.
 *
 * The HIP product (webgpu-pathtracer_amd/csrc/pt_devmath.h) implements the same
 * specification independently; tests/test_math_parity.py holds the two bit-equal
 * and checks both against float64 libm within a few ULP.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and cannot be run
 * in this environment, so nothing in the reference pins these
 * choices.
 */
#ifndef PT_ORACLE_MATH_H
#define PT_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t om_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float om_float(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

/* 2^n for n in [-126, 127] built from bits. */
static inline float om_pow2i(int n) { return om_float((uint32_t)(n + 127) << 23); }

/* z * 2^n with a single rounding, n in [-150, 254]. */
static inline float om_ldexp(float z, int n)
{
    if (n > 127) {
        z = z * om_pow2i(127);
        n -= 127;
        if (n > 127) n = 127;
        return z * om_pow2i(n);
    }
    if (n < -126) {
        n += 24;
        if (n < -126) n = -126;
        return (z * om_pow2i(n)) * om_pow2i(-24);
    }
    return z * om_pow2i(n);
}

/* ---- sin / cos: Cody-Waite reduction by pi/2, Cephes sinf/cosf polynomials ---- */
static inline void om_sincos(float x, float *s_out, float *c_out)
{
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float P1 = 1.5703125f;                  /* pi/2 split in three */
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188e-8f;
    float q = rintf(x * TWO_OVER_PI);
    float r = fmaf(-q, P1, x);
    r = fmaf(-q, P2, r);
    r = fmaf(-q, P3, r);
    int n = (int)q;
    float z = r * r;
    /* sin(r) = r + r*z*(S1 + z*(S2 + z*S3)) */
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float s = fmaf(ps * z, r, r);
    /* cos(r) = 1 - z/2 + z*z*(C1 + z*(C2 + z*C3)) */
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float c = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    float rs, rc;
    switch (n & 3) {
    case 0: rs = s; rc = c; break;
    case 1: rs = c; rc = -s; break;
    case 2: rs = -s; rc = -c; break;
    default: rs = -c; rc = s; break;
    }
    *s_out = rs;
    *c_out = rc;
}
static inline float om_sin(float x) { float s, c; om_sincos(x, &s, &c); return s; }
static inline float om_cos(float x) { float s, c; om_sincos(x, &s, &c); return c; }
/* tan = sin / cos with a correctly rounded division. */
static inline float om_tan(float x) { float s, c; om_sincos(x, &s, &c); return s / c; }

/* ---- log: Cephes logf ---- */
static inline float om_log(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return om_float(0x7fc00000u);
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u = om_bits(x);
    if ((u >> 23) == 0) {            /* denormal: scale by 2^25 first */
        x = x * 33554432.0f;
        u = om_bits(x);
        e = -25;
    }
    e += (int)(u >> 23) - 126;       /* x = m * 2^e, m in [0.5, 1) */
    float m = om_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* ---- exp: Cephes expf ---- */
static inline float om_exp(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.972077083991796f) return 0.0f;
    float k = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    float r = fmaf(-k, 0.693359375f, x);
    r = fmaf(-k, -2.12194440e-4f, r);
    float z = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, z, r) + 1.0f;
    return om_ldexp(y, (int)k);
}

/* pow(x, y) = exp(y * log(x)); only meaningful for x >= 0 (WGSL leaves x < 0 undefined). */
static inline float om_pow(float x, float y) { return om_exp(y * om_log(x)); }

/* ---- atan / atan2: Cephes atanf ---- */
static inline float om_atan(float x)
{
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) {         /* tan(3 pi / 8) */
        y = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi / 8) */
        y = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return sign * y;
}

static inline float om_atan2(float y, float x)
{
    const float PI_F = 3.14159265358979323846f;
    const float PIO2_F = 1.5707963267948966f;
    if (x != x || y != y) return om_float(0x7fc00000u);
    if (x == 0.0f) {
        if (y > 0.0f) return PIO2_F;
        if (y < 0.0f) return -PIO2_F;
        return 0.0f;
    }
    float a = om_atan(y / x);
    if (x > 0.0f) return a;
    if (y >= 0.0f) return a + PI_F;
    return a - PI_F;
}

/* ---- asin: Cephes asinf ---- */
static inline float om_asin(float x)
{
    float sign = 1.0f;
    float a = x;
    if (x < 0.0f) { sign = -1.0f; a = -x; }
    if (a != a) return a;
    if (a > 1.0f) return om_float(0x7fc00000u);
    if (a < 1.0e-4f) return x;
    int flag = 0;
    float z, t;
    if (a > 0.5f) {
        z = 0.5f * (1.0f - a);
        t = sqrtf(z);
        flag = 1;
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

/* ---- fp16 storage emulation (rgba16float textures), round-to-nearest-even ---- */
static inline uint16_t om_f32_to_f16_bits(float f)
{
    uint32_t x = om_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) {                       /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x0200u : 0u));
    }
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);   /* >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                  /* <= 2^-25 -> 0 */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    int shift;
    uint32_t he;
    if (e < -14) { shift = 13 + (-14 - e); he = 0; }    /* subnormal half */
    else { shift = 13; he = (uint32_t)(e + 15); }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q += 1;
    uint32_t h;
    if (he == 0) h = q;                        /* q may carry into the exponent: fine */
    else h = ((he - 1) << 10) + q;             /* q includes the hidden bit (0x400) */
    return (uint16_t)(sign | h);
}

static inline float om_f16_bits_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return om_float(sign);
        float v = (float)m * 5.9604644775390625e-8f;     /* m * 2^-24, exact */
        return (sign ? -v : v);
    }
    if (e == 31) return om_float(sign | 0x7f800000u | (m << 13));
    return om_float(sign | ((e + 112u) << 23) | (m << 13));
}

static inline float om_round_f16(float f) { return om_f16_bits_to_f32(om_f32_to_f16_bits(f)); }

#endif
