// pt_kernels.hip -- gfx950 kernels for the per-pixel ray-trace path.
//
// Hand-written HIP for MI355X; follows the reference's shaders function by function
// (citations are to /src/passes/shaders/*.wgsl of umar-ahmed/webgpu-pathtracer) under
// the arithmetic pinned in DESIGN.md.  Build with -ffp-contract=off, no fast-math.
//
// Kernel inventory
//   k_raytrace<FUSE, VARIANT>  raytrace.wgsl computeMain (+ accumulate.wgsl when FUSE)
//       one wave64 = one 8x8 pixel tile (the reference's @workgroup_size(8,8));
//       traversal stack in LDS ([depth][lane], conflict free);
//       VARIANT 1 walks the uploaded 48-B node records exactly like the WGSL,
//       VARIANT 2 walks 64-B node packets (both child boxes in one line) and 48-B
//       triangle packets -- same tests in the same order, a third of the loads.
//       The on-device references of the parity tests (kernel variants 1 and 2), and the fall-back
//       for launches beyond the state-machine kernel's packing limits.
//   k_raytrace_sm<DEFER, CULL, WIDE, FILT, YMAX, DIAG, TOPLDS, LITE, CW>
//       the shipped kernel: persistent one-wave workgroups, per-lane state machine with node /
//       triangle / service steps (the service step split into a hit-shading and a miss / path
//       group), (frame slot, tile) jobs from a self-cleaning queue; one packed per-lane state for
//       every launch (texel index, frame slot | bounce; a multi-sample frame's sum lives in the texel);
//       !DEFER = in-order walk (variant 4), DEFER = leaves parked and tested in steps of their own
//       (variant 7: exactly the reference's tests), CULL = DEFER + exact-image distance culling,
//       children near first (variant 9), WIDE = CULL on 4-ary wide packets (variant 10), FILT = the
//       filtered slab test (11), YMAX = the one-axis culling condition (12), CW = compressed wide
//       packets + 64-B triangle records with the leaf's box (13: the default where the tree admits it);
//       DIAG = false: the lean build every ordinary launch runs (80 - 96 VGPRs, five waves per SIMD),
//       true: the diagnostic twin (step statistics, clock stamps, step-voting options at run time);
//       TOPLDS / LITE and k_raytrace_persistent (variant 3): experiment builds only
//   k_rt_service_setup         the launch-invariant scalars of the service step, one block per launch
//   k_accumulate[_batch]       accumulate.wgsl computeMain (one frame / an ordered batch of frames)
//   k_fullscreen[_setup]       fullscreen.wgsl fragmentMain (de-noise + tone-map) and its tap table
//   k_pack_vertices, k_patch_cull   helpers of the context's cull analysis
//   k_debug_intersect/_math    component probes for the parity tests
#include "pt_kernels.h"
#include "pt_devmath.h"
#include <hip/hip_fp16.h>

namespace pt {

// raytrace.wgsl:1-8
#define PT_SEED 123456789u
#define PT_TWOPI 6.28318530718f
#define PT_INVPI 0.31830988618f
#define PT_INVTWOPI 0.15915494309f
#define PT_INF 1e20f
#define PT_EPSILON 1e-6f
#define PT_MAX_STACK 64

#define PT_REF_LEAF pt::REF_LEAF
#define PT_REF_NONE pt::REF_NONE

struct f3 { float x, y, z; };

using ptm::div_pre;
using ptm::rcp_exact;

// An opaque use-and-redefinition of a 16-byte value: keeps the load that produced it ONE dwordx4 (left alone, the compiler narrows
// and splits such loads to feed packed operations).  WHOLE: as one 128-bit operand, so that its four dwords are not tied to four
// separate registers -- which costs a copy per dword out of the load's register tuple; !WHOLE: four 32-bit operands (the registers
// may part ways: what the instantiations without a register to spare for whole tuples keep -- they spill with WHOLE).
typedef float pt_keep4 __attribute__((ext_vector_type(4)));
template <bool WHOLE>
PT_DEV void keep16(float4 &v)
{
    if constexpr (WHOLE) {
        pt_keep4 t = { v.x, v.y, v.z, v.w };
        asm volatile("" : "+v"(t));
        v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
    } else {
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    }
}
PT_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 xyz(const float4 &v) { return F3(v.x, v.y, v.z); }
PT_DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
PT_DEV f3 neg(f3 a) { return F3(-a.x, -a.y, -a.z); }
// pinned: dot summed left to right, no contraction
PT_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_DEV f3 cross(f3 a, f3 b)
{
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// pinned: normalize(v) = v / sqrt(dot(v, v)), correctly rounded sqrt and divisions.  The
// compiler's IEEE expansions cost 14 + 3 x 11 instructions; the same bits come from
// ptm::sqrt_exact, one exact reciprocal of the length and ptm::div_pre per component whenever
// no intermediate can underflow (length within [2^-40, 2^40], components 0 or >= 2^-80), which
// is checked; anything else takes the plain operations.
PT_DEV f3 normalize(f3 a)
{
    const float l = ptm::sqrt_exact(dot(a, a));
    const uint32_t lo = 0x17800000u - 1u;                         // 2^-80
    const bool comps = ((__float_as_uint(a.x) & 0x7fffffffu) - 1u >= lo) && ((__float_as_uint(a.y) & 0x7fffffffu) - 1u >= lo) &&
                       ((__float_as_uint(a.z) & 0x7fffffffu) - 1u >= lo);      // 0 wraps to 0xffffffff: passes
    if (comps && l >= 9.094947017729282e-13f && l <= 1.099511627776e12f) {
        const float y0 = __builtin_amdgcn_rcpf(l);
        const float y = fmaf(fmaf(-l, y0, 1.0f), y0, y0);          // RN(1/l), see ptm::rcp_exact
        return F3(div_pre(a.x, l, y), div_pre(a.y, l, y), div_pre(a.z, l, y));
    }
    return F3(a.x / l, a.y / l, a.z / l);
}
// pinned: mix(a, b, t) = a * (1 - t) + b * t
PT_DEV float mix1(float a, float b, float t) { return a * (1.0f - t) + b * t; }
PT_DEV f3 mix(f3 a, f3 b, float t) { return F3(mix1(a.x, b.x, t), mix1(a.y, b.y, t), mix1(a.z, b.z, t)); }
// pinned: reflect(i, n) = i - (2 * dot(n, i)) * n
PT_DEV f3 reflect(f3 i, f3 n)
{
    const float k = 2.0f * dot(n, i);
    return i - n * k;
}
PT_DEV float clamp1(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

struct Counters {
    uint32_t rays, box, tri, hit, miss, overflow, pixels;
    uint32_t slow;      // path segments that took the plain-division fallback of the fast slab test
};

struct Best {
    float t, u, v;
    int32_t tri;     // -1 = no hit
};

// ---------------------------------------------------------------------------------
// raytrace.wgsl:118-152 -- slab test, true divisions, early outs folded into one
// result (tmin only grows and tmax only shrinks, so "tmin > tmax after some axis"
// is the same predicate as "tmin > tmax after the last axis").
// ---------------------------------------------------------------------------------
PT_DEV bool ray_aabb(const f3 &o, const f3 &d, float mnx, float mny, float mnz,
                     float mxx, float mxy, float mxz)
{
    float tmin = -PT_INF, tmax = PT_INF;
    bool ok = true;
#define PT_SLAB(DIR, ORG, MN, MX)                                   \
    if (fabsf(DIR) < PT_EPSILON) {                                  \
        if (ORG < MN || ORG > MX) ok = false;                       \
    } else {                                                        \
        const float t1 = (MN - ORG) / DIR;                          \
        const float t2 = (MX - ORG) / DIR;                          \
        tmin = fmaxf(tmin, fminf(t1, t2));                          \
        tmax = fminf(tmax, fmaxf(t1, t2));                          \
        if (tmin > tmax) ok = false;                                \
    }
    PT_SLAB(d.x, o.x, mnx, mxx)
    PT_SLAB(d.y, o.y, mny, mxy)
    PT_SLAB(d.z, o.z, mnz, mxz)
#undef PT_SLAB
    return ok && (tmax >= fmaxf(0.0f, tmin));
}

// ---------------------------------------------------------------------------------
// The same slab test with the twelve IEEE divisions of a node's two boxes replaced by
// a per-ray reciprocal.  All six quotients of a box share the ray's three divisors, so
// y = RN(1/d) is formed once per path segment (one true division per axis) and each
// quotient is recovered EXACTLY as RN(n/d) by one Newton correction with an exact fma
// residual:
//     q0 = n*y;  q1 = fma(fma(-d, q0, n), y, q0)
// 3 VALU ops instead of the 11 of the hardware division expansion.  That q1 == RN(n/d)
// for every binary32 n, d is established by exhaustion: profiles/div_proof.hip checks all
// 2^23 x 2^23 significand pairs on the device (7.04e13 pairs, 0 mismatches -- log in
// profiles/r01_g_div_proof_exhaustive.log; q0 alone is wrong for 27 % of them), and every
// operation is round-to-nearest, so the identity carries over to all signs and exponents
// as long as nothing under- or overflows.  That proviso is checked per ray
// (RayPre::slow: |d| within [1e-6, 2^20], origin components 0 or within [2^-70, 2^60])
// and per box (a packet flag set at upload for the rare boxes with a non-zero coordinate
// outside [2^-70, 2^60], e.g. the 1e-33 residues three.js's SphereGeometry leaves at the
// poles); a test that fails it takes ray_aabb() above.  tests/test_gpu_parity.py holds
// this kernel bit-identical to the plain-division kernels; tests/test_math_oracle.py
// samples the identity over the admitted exponent range on the CPU.
// ---------------------------------------------------------------------------------
struct RayPre {
    float ix, iy, iz;        // RN(1/d) per axis
    uint32_t flags;          // bit3: this ray takes the plain-division test
};

PT_DEV bool safe_magnitude(float v)
{
    const float a = fabsf(v);
    return v == 0.0f || (a >= 8.470329472543003e-22f && a <= 1.152921504606847e18f);   // [2^-70, 2^60]
}

PT_DEV RayPre ray_prepare(const f3 &o, const f3 &d, uint32_t scene_flags)
{
    RayPre p;
    (void)scene_flags;
    // A parallel axis (|d| < EPSILON, raytrace.wgsl:129-133) is rare; such rays take the
    // plain test, which keeps selects out of the fast path below.
    const bool px = fabsf(d.x) < PT_EPSILON, py = fabsf(d.y) < PT_EPSILON, pz = fabsf(d.z) < PT_EPSILON;
    p.ix = rcp_exact(d.x);
    p.iy = rcp_exact(d.y);
    p.iz = rcp_exact(d.z);
    // range guards (no residual may under- or overflow)
    const bool bad_x = fabsf(d.x) > 1048576.0f, bad_y = fabsf(d.y) > 1048576.0f, bad_z = fabsf(d.z) > 1048576.0f;
    const bool slow = px || py || pz || bad_x || bad_y || bad_z || !safe_magnitude(o.x) ||
                      !safe_magnitude(o.y) || !safe_magnitude(o.z) || !(d.x == d.x) || !(d.y == d.y) || !(d.z == d.z);
    p.flags = slow ? 8u : 0u;
    return p;
}

PT_DEV bool ray_aabb_pre(const f3 &o, const f3 &d, const RayPre &p, bool box_unsafe, float mnx, float mny, float mnz,
                         float mxx, float mxy, float mxz)
{
    if ((p.flags & 8u) || box_unsafe) return ray_aabb(o, d, mnx, mny, mnz, mxx, mxy, mxz);
    const float ax = div_pre(mnx - o.x, d.x, p.ix), bx = div_pre(mxx - o.x, d.x, p.ix);
    const float ay = div_pre(mny - o.y, d.y, p.iy), by = div_pre(mxy - o.y, d.y, p.iy);
    const float az = div_pre(mnz - o.z, d.z, p.iz), bz = div_pre(mxz - o.z, d.z, p.iz);
    const float tmin = fmaxf(fmaxf(fmaxf(-PT_INF, fminf(ax, bx)), fminf(ay, by)), fminf(az, bz));
    const float tmax = fminf(fminf(fminf(PT_INF, fmaxf(ax, bx)), fmaxf(ay, by)), fmaxf(az, bz));
    return !(tmin > tmax) && (tmax >= fmaxf(0.0f, tmin));
}

// ray_aabb_pre() for a ray and a box that both passed the guards (no fallback inside)
PT_DEV bool ray_aabb_fast(const f3 &o, const f3 &d, const RayPre &p, float mnx, float mny, float mnz,
                          float mxx, float mxy, float mxz)
{
    const float ax = div_pre(mnx - o.x, d.x, p.ix), bx = div_pre(mxx - o.x, d.x, p.ix);
    const float ay = div_pre(mny - o.y, d.y, p.iy), by = div_pre(mxy - o.y, d.y, p.iy);
    const float az = div_pre(mnz - o.z, d.z, p.iz), bz = div_pre(mxz - o.z, d.z, p.iz);
    const float tmin = fmaxf(fmaxf(fmaxf(-PT_INF, fminf(ax, bx)), fminf(ay, by)), fminf(az, bz));
    const float tmax = fminf(fminf(fminf(PT_INF, fmaxf(ax, bx)), fmaxf(ay, by)), fmaxf(az, bz));
    return !(tmin > tmax) && (tmax >= fmaxf(0.0f, tmin));
}

// ray_aabb_fast() that also hands out the per-axis entry distances min(t1, t2) (CULL walk: the
// distance bound and the near-first ordering; the hit predicate is the same expression)
PT_DEV bool ray_aabb_fast_t(const f3 &o, const f3 &d, const RayPre &p, float mnx, float mny, float mnz,
                            float mxx, float mxy, float mxz, f3 &tnear)
{
    const float ax = div_pre(mnx - o.x, d.x, p.ix), bx = div_pre(mxx - o.x, d.x, p.ix);
    const float ay = div_pre(mny - o.y, d.y, p.iy), by = div_pre(mxy - o.y, d.y, p.iy);
    const float az = div_pre(mnz - o.z, d.z, p.iz), bz = div_pre(mxz - o.z, d.z, p.iz);
    tnear = F3(fminf(ax, bx), fminf(ay, by), fminf(az, bz));
    const float tmin = fmaxf(fmaxf(fmaxf(-PT_INF, tnear.x), tnear.y), tnear.z);
    const float tmax = fminf(fminf(fminf(PT_INF, fmaxf(ax, bx)), fmaxf(ay, by)), fmaxf(az, bz));
    return !(tmin > tmax) && (tmax >= fmaxf(0.0f, tmin));
}

// ---------------------------------------------------------------------------------
// Filtered slab test (the WIDE walk's fast path): decide with approximate quotients, run the exact
// test only where the approximation cannot decide.
//
// q0 = RN(n * y), y = RN(1/d), is div_pre()'s first operation.  With r = n/d (real): y = (1/d)(1+e1),
// q0 = r (1+e1)(1+e2), q1 = RN(n/d) = r (1+e3), |e_i| <= u = 2^-24, so |q0 - q1| <= eps |q1| with
// eps = 3.01 u; q0 and q1 have the same sign and q0 == 0 exactly when q1 == 0 (n == 0: the guards
// that admit a ray and a box to the fast path exclude under- and overflow of n * y).  x -> x +- eps |x|
// are increasing maps and min / max commute with increasing maps, so the same bound carries over to
// any min / max combination of the six quotients (constants such as +-PT_INF are their own images):
//     |tmin~ - tmin| <= eps |tmin|,   |tmax~ - tmax| <= eps |tmax|,   sign(tmax~) == sign(tmax)
// for tmin~ = max3 of the per-axis minima and tmax~ = min(PT_INF, per-axis maxima) of the q0's.
// The reference's predicate (raytrace.wgsl:145-151) is  !(tmin > tmax) && tmax >= max(0, tmin),  which
// without NaNs (none on this path) is  tmin <= tmax && tmax >= 0;  clamping tmin at -PT_INF from below
// cannot change it (a tmin below -PT_INF lies below every tmax that is >= 0).  With D~ = RN(tmin~ - tmax~)
// and M = max(|tmin~|, |tmax~|):
//     |(tmin~ - tmax~) - (tmin - tmax)| <= eps (|tmin| + |tmax|) <= 2 eps / (1 - eps) M  <  6.03 u M,
// so once |D~| > 2^-21 M (= 8 u M; the fma that forms |D~| - 2^-21 M rounds a positive real to a value
// >= 0 and D~ itself carries one rounding) the sign of tmin - tmax is the sign of D~, and the predicate
// is  D~ < 0 && tmax~ >= 0  bit for bit.  Otherwise the box is UNDECIDED (slab_margin() <= 0) and the
// caller runs ray_aabb_fast() on it.  profiles/slab_filter_proof.hip replays this against the exact test.
// The per-axis entry distances are handed out for the culling bound and the sort keys, which only
// need them to within the 4.01 u the proof of DESIGN.md 3a budgets (here: 3.01 u).
// ---------------------------------------------------------------------------------
#define PT_SLAB_BAND 4.76837158203125e-07f      // 2^-21
PT_DEV void slab_q0(const f3 &o, const RayPre &p, float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                    f3 &tnear, float &tfar)
{
    const float ax = (mnx - o.x) * p.ix, bx = (mxx - o.x) * p.ix;
    const float ay = (mny - o.y) * p.iy, by = (mxy - o.y) * p.iy;
    const float az = (mnz - o.z) * p.iz, bz = (mxz - o.z) * p.iz;
    tnear = F3(fminf(ax, bx), fminf(ay, by), fminf(az, bz));
    tfar = fminf(fminf(fminf(PT_INF, fmaxf(ax, bx)), fmaxf(ay, by)), fmaxf(az, bz));
}
// > 0: the approximate quantities decide the box;  <= 0 (or NaN): undecided
PT_DEV float slab_margin(float tmin, float tfar)
{
    return fmaf(-PT_SLAB_BAND, fmaxf(fabsf(tmin), fabsf(tfar)), fabsf(tmin - tfar));
}
// the predicate for a decided box
PT_DEV bool slab_hit(float tmin, float tfar) { return fmaxf(tmin - tfar, -tfar) <= 0.0f; }

// ---------------------------------------------------------------------------------
// Compressed wide packets (kernel variant 13): a test that NEVER rejects a box the reference's exact test would pass.
//
// The packet holds boxes on an 8-bit grid, rounded outward by >= one whole cell c_i = 2^e_i per side (pt_kernels.h, CWidePacket):
// decoded plane p' = o_i + c_i q lies >= c_i outside the child's plane p.  The kernel forms each quotient as ONE fma,
//     t~ = fma(q, B_i, A_i),   B_i = c_i * RN(1/d_i) (exact: a power of two),   A_i = RN(RN(o_i - O_i) * RN(1/d_i))     (O = ray origin),
// whose value differs from the real quotient t' = (p' - O_i) / d_i by at most 3u |t'| + 2.01u |A_i|, and |A_i| <= |t'| + 255 c_i / |d_i|:
//     |t~ - t'|  <=  5.02u |t'|  +  3.1e-5 * c_i / |d_i|.
// The second term is a 3e-5-th of the cell the plane was moved outward by (c_i / |d_i| in t), so t~ is, up to a RELATIVE error of
// 5.02u, the real quotient of a plane that still lies >= 0.99996 c_i outside the child's: of a box E that contains the child's box
// B.  Let the reference's fp32 test pass for B (raytrace.wgsl:118-152: tmin <= tmax and tmax >= 0, its quotients correctly
// rounded: relative error <= 1.5u each, signs exact).  Then in real arithmetic tn(B) - tf(B) <= 3u max(|tn|, |tf|) and tf(B) >= 0; E
// contains B, so tn(E) <= tn(B), tf(E) >= tf(B) + 0.99996 c_i / |d_i| > 0; and for the computed values (min / max commute with the
// increasing maps x -> x (1 +- 5.02u)):  tmin~ - tfar~ <= (3 + 10.04) u M < 2^-20 M  with M = max(|tmin~|, |tfar~|), and tfar~ > 0
// (its real value exceeds c_i / |d_i| - ..., far above its error).  So rejecting only when
//     tmin~ - tfar~ > 2^-20 M   or   tfar~ < 0
// never rejects a box the reference passes.  False accepts cost work, never a bit: the reference reaches a leaf iff its LEAF box
// passes (every box nested: a leaf that passes has every ancestor pass, the monotonicity behind the wide collapse), and that test
// is made exactly, on the leaf's own box, in the triangle step (leaf_box_hit).  The per-axis entry distances handed to the culling
// bound are <= the child's own (E contains B) up to 5.02u, inside what DESIGN.md 3a budgets once its 4.01u is read as 5.02u -- the
// bound's factor (1 + 2^-20) has room for 16u.
// ---------------------------------------------------------------------------------
#define PT_CW_BAND 9.5367431640625e-07f      // 2^-20
// (round 6) The first condition in the form  tmin~ (1 - 2^-20) > tfar~ , ONE fma whose sign is the exact sign of tmin~ (1 - 2^-20) - tfar~
// (1 - 2^-20 is a binary32 number; a single rounding keeps the sign of a non-zero real, and a result that underflows to zero reads "not
// greater": the conservative side).  It is the same condition wherever it matters: a box with tfar~ < 0 is rejected by the second condition
// whatever the first says; with tfar~ >= 0 the product can only exceed tfar~ when tmin~ > tfar~ >= 0, i.e. M = tmin~, and then
// tmin~ (1 - 2^-20) > tfar~  <=>  tmin~ - tfar~ > 2^-20 M.  Two instructions (a max of two magnitudes, a subtraction) fewer per box than
// the form with M spelled out, and no rounding of its own between the quotients and the decision.
#define PT_CW_SHRINK 0.99999904632568359375f      // 1 - 2^-20
PT_DEV bool cwide_hit(float tmin, float tfar)
{
    return !(fmaf(tmin, PT_CW_SHRINK, -tfar) > 0.0f) && !(tfar < 0.0f);
}

// the reference's test of a LEAF's box (raytrace.wgsl:186, 194) for the compressed-wide walk's triangle step: the filtered test,
// the exact one where that cannot decide, the plain-division one for the rays that take it everywhere
PT_DEV bool leaf_box_hit(const f3 &o, const f3 &d, const RayPre &p, bool box_unsafe, const f3 &mn, const f3 &mx)
{
    if ((p.flags & 8u) || box_unsafe) return ray_aabb(o, d, mn.x, mn.y, mn.z, mx.x, mx.y, mx.z);
    f3 tn;
    float tf;
    slab_q0(o, p, mn.x, mn.y, mn.z, mx.x, mx.y, mx.z, tn, tf);
    const float key = fmaxf(fmaxf(tn.x, tn.y), tn.z);
    bool h = slab_hit(key, tf);
    if (!(slab_margin(key, tf) > 0.0f)) h = ray_aabb_fast(o, d, p, mn.x, mn.y, mn.z, mx.x, mx.y, mx.z);
    return h;
}

// raytrace.wgsl:78-116 -- Moller-Trumbore, two-sided.  Returns hit and (t, u, v);
// position and normal are formed once, for the closest hit, by finish_hit().
PT_DEV bool ray_triangle(const f3 &o, const f3 &d, const f3 &a, const f3 &b, const f3 &c,
                         float &t_out, float &u_out, float &v_out)
{
    const f3 edge1 = b - a;
    const f3 edge2 = c - a;
    const f3 h = cross(d, edge2);
    const float det = dot(edge1, h);
    if (det > -PT_EPSILON && det < PT_EPSILON) return false;
    const float f = rcp_exact(det);
    const f3 s = o - a;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = f * dot(edge2, q);
    if (t > PT_EPSILON) {
        t_out = t; u_out = u; v_out = v;
        return true;
    }
    return false;
}

// The same test on a triangle PACKET: vertex a and the edges edge1 = fl(b - a), edge2 = fl(c - a), formed at upload (TriPacket) --
// the values the first two lines above compute, so every later operation sees the same operands.
PT_DEV bool ray_triangle_e(const f3 &o, const f3 &d, const f3 &a, const f3 &edge1, const f3 &edge2,
                           float &t_out, float &u_out, float &v_out)
{
    const f3 h = cross(d, edge2);
    const float det = dot(edge1, h);
    if (det > -PT_EPSILON && det < PT_EPSILON) return false;
    const float f = rcp_exact(det);
    const f3 s = o - a;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = f * dot(edge2, q);
    if (t > PT_EPSILON) {
        t_out = t; u_out = u; v_out = v;
        return true;
    }
    return false;
}
// ... and without early returns (the state-machine kernel's triangle step: lanes leave a wave-wide test at different points
// anyway, and every return is a branch + lane-mask bookkeeping): all quantities are formed, the conditions are the ones
// above in the same sense for NaNs, and t / u / v are only meaningful when it returns true.
PT_DEV bool ray_triangle_flat_e(const f3 &o, const f3 &d, const f3 &a, const f3 &edge1, const f3 &edge2,
                                float &t, float &u, float &v)
{
    const f3 h = cross(d, edge2);
    const float det = dot(edge1, h);
    const float f = rcp_exact(det);
    const f3 s = o - a;
    u = f * dot(s, h);
    const f3 q = cross(s, edge1);
    v = f * dot(d, q);
    t = f * dot(edge2, q);
    const bool degenerate = det > -PT_EPSILON && det < PT_EPSILON;
    const bool out_u = u < 0.0f || u > 1.0f;
    const bool out_v = v < 0.0f || u + v > 1.0f;
    return !degenerate && !out_u && !out_v && t > PT_EPSILON;
}

// raytrace.wgsl:154-203 (+ :205-211) on the uploaded records, as written: root box
// first; pop; leaf -> its triangle, strict '<' keeps the first of equal t; internal ->
// test left then right child box, push in that order (right is popped first); abort
// with best-so-far when the stack holds 64 entries at the top of the loop.
PT_DEV void traverse_generic(const SceneRefs &sc, const f3 &o, const f3 &d,
                             uint32_t *stack /* LDS, stride 64 */, Best &best, Counters &cnt)
{
    best.t = PT_INF; best.u = 0.0f; best.v = 0.0f; best.tri = -1;
    cnt.rays++;
    if (sc.nnodes == 0) return;
    {
        const float4 n0 = sc.nodes[0], n1 = sc.nodes[1];
        cnt.box++;
        if (!ray_aabb(o, d, n0.x, n0.y, n0.z, n1.x, n1.y, n1.z)) return;
    }
    int sp = 0;
    stack[0] = 0u;
    sp = 1;
    while (sp > 0) {
        if (sp >= PT_MAX_STACK) { cnt.overflow++; return; }
        sp--;
        const uint32_t cur = stack[sp * 64];
        const float4 c1 = sc.nodes[(size_t)cur * 3 + 1];
        const float4 c2 = sc.nodes[(size_t)cur * 3 + 2];
        const int32_t is_leaf = __float_as_int(c1.w);
        if (is_leaf == 1) {
            const uint32_t ti = (uint32_t)__float_as_int(c2.z);
            const float4 pa = sc.tris[(size_t)ti * 7 + 0];
            const float4 pb = sc.tris[(size_t)ti * 7 + 1];
            const float4 pc = sc.tris[(size_t)ti * 7 + 2];
            cnt.tri++;
            float t, u, v;
            if (ray_triangle(o, d, xyz(pa), xyz(pb), xyz(pc), t, u, v) && t < best.t) {
                best.t = t; best.u = u; best.v = v; best.tri = (int32_t)ti;
            }
        } else {
            const int32_t left = __float_as_int(c2.x), right = __float_as_int(c2.y);
            if (left >= 0) {
                const float4 l0 = sc.nodes[(size_t)left * 3], l1 = sc.nodes[(size_t)left * 3 + 1];
                cnt.box++;
                if (ray_aabb(o, d, l0.x, l0.y, l0.z, l1.x, l1.y, l1.z)) { stack[sp * 64] = (uint32_t)left; sp++; }
            }
            if (right >= 0) {
                const float4 r0 = sc.nodes[(size_t)right * 3], r1 = sc.nodes[(size_t)right * 3 + 1];
                cnt.box++;
                if (ray_aabb(o, d, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z)) { stack[sp * 64] = (uint32_t)right; sp++; }
            }
        }
    }
}

// The same walk on node / triangle packets: identical tests in identical order, the
// stack carries child references instead of node indices.
PT_DEV void traverse_packets(const SceneRefs &sc, const f3 &o, const f3 &d,
                             uint32_t *stack, Best &best, Counters &cnt)
{
    best.t = PT_INF; best.u = 0.0f; best.v = 0.0f; best.tri = -1;
    cnt.rays++;
    if (sc.nnodes == 0) return;
    {
        const float4 n0 = sc.nodes[0], n1 = sc.nodes[1];
        cnt.box++;
        if (!ray_aabb(o, d, n0.x, n0.y, n0.z, n1.x, n1.y, n1.z)) return;
    }
    int sp = 1;
    stack[0] = sc.root_ref;
    while (sp > 0) {
        if (sp >= PT_MAX_STACK) { cnt.overflow++; return; }
        sp--;
        const uint32_t ref = stack[sp * 64];
        if (ref & PT_REF_LEAF) {
            const uint32_t ti = ref & 0x7fffffffu;
            const float4 pa = sc.tripk[(size_t)ti * 3 + 0];
            const float4 pb = sc.tripk[(size_t)ti * 3 + 1];
            const float4 pc = sc.tripk[(size_t)ti * 3 + 2];
            cnt.tri++;
            float t, u, v;
            if (ray_triangle_e(o, d, xyz(pa), xyz(pb), xyz(pc), t, u, v) && t < best.t) {
                best.t = t; best.u = u; best.v = v; best.tri = (int32_t)ti;
            }
        } else {
            const float4 p0 = sc.packets[(size_t)ref * 4 + 0];   // lmin.xyz lmax.x
            const float4 p1 = sc.packets[(size_t)ref * 4 + 1];   // lmax.yz rmin.xy
            const float4 p2 = sc.packets[(size_t)ref * 4 + 2];   // rmin.z rmax.xyz
            const float4 p3 = sc.packets[(size_t)ref * 4 + 3];   // lref rref
            const uint32_t lref = __float_as_uint(p3.x), rref = __float_as_uint(p3.y);
            if (lref != PT_REF_NONE) {
                cnt.box++;
                if (ray_aabb(o, d, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y)) { stack[sp * 64] = lref; sp++; }
            }
            if (rref != PT_REF_NONE) {
                cnt.box++;
                if (ray_aabb(o, d, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w)) { stack[sp * 64] = rref; sp++; }
            }
        }
    }
}

// The packet walk with the prepared-reciprocal slab test (what the state-machine kernel
// executes per lane), as one call: used by the probe kernel.
PT_DEV void traverse_packets_pre(const SceneRefs &sc, const f3 &o, const f3 &d,
                                 uint32_t *stack, Best &best, Counters &cnt)
{
    best.t = PT_INF; best.u = 0.0f; best.v = 0.0f; best.tri = -1;
    cnt.rays++;
    if (sc.nnodes == 0) return;
    const RayPre pre = ray_prepare(o, d, sc.flags);
    {
        const float4 n0 = sc.nodes[0], n1 = sc.nodes[1];
        cnt.box++;
        if (!ray_aabb_pre(o, d, pre, (sc.flags & 1u) == 0u, n0.x, n0.y, n0.z, n1.x, n1.y, n1.z)) return;
    }
    int sp = 1;
    stack[0] = sc.root_ref;
    while (sp > 0) {
        if (sp >= PT_MAX_STACK) { cnt.overflow++; return; }
        sp--;
        const uint32_t ref = stack[sp * 64];
        if (ref & PT_REF_LEAF) {
            const uint32_t ti = ref & 0x7fffffffu;
            const float4 pa = sc.tripk[(size_t)ti * 3 + 0];
            const float4 pb = sc.tripk[(size_t)ti * 3 + 1];
            const float4 pc = sc.tripk[(size_t)ti * 3 + 2];
            cnt.tri++;
            float t, u, v;
            if (ray_triangle_e(o, d, xyz(pa), xyz(pb), xyz(pc), t, u, v) && t < best.t) {
                best.t = t; best.u = u; best.v = v; best.tri = (int32_t)ti;
            }
        } else {
            const float4 p0 = sc.packets[(size_t)ref * 4 + 0];
            const float4 p1 = sc.packets[(size_t)ref * 4 + 1];
            const float4 p2 = sc.packets[(size_t)ref * 4 + 2];
            const float4 p3 = sc.packets[(size_t)ref * 4 + 3];
            const uint32_t lref = __float_as_uint(p3.x), rref = __float_as_uint(p3.y);
            if (lref != PT_REF_NONE) {
                cnt.box++;
                if (ray_aabb_pre(o, d, pre, (__float_as_uint(p3.z) & 1u) != 0u, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y)) { stack[sp * 64] = lref; sp++; }
            }
            if (rref != PT_REF_NONE) {
                cnt.box++;
                if (ray_aabb_pre(o, d, pre, (__float_as_uint(p3.z) & 2u) != 0u, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w)) { stack[sp * 64] = rref; sp++; }
            }
        }
    }
}

template <int VARIANT>
PT_DEV void traverse(const SceneRefs &sc, const f3 &o, const f3 &d, uint32_t *stack, Best &best,
                     Counters &cnt)
{
    if (VARIANT == 3) traverse_packets_pre(sc, o, d, stack, best, cnt);
    else if (VARIANT == 2) traverse_packets(sc, o, d, stack, best, cnt);
    else traverse_generic(sc, o, d, stack, best, cnt);
}

// Position, shading normal and material of the closest hit (raytrace.wgsl:105-112).
PT_DEV void finish_hit(const SceneRefs &sc, const f3 &o, const f3 &d, const Best &best,
                       f3 &position, f3 &normal, int32_t &material)
{
    const float4 q3 = sc.tris[(size_t)best.tri * 7 + 3];
    const float4 q4 = sc.tris[(size_t)best.tri * 7 + 4];
    const float4 q5 = sc.tris[(size_t)best.tri * 7 + 5];
    const float w = 1.0f - best.u - best.v;
    position = o + d * best.t;
    normal = normalize((xyz(q3) * w + xyz(q4) * best.u) + xyz(q5) * best.v);
    material = __float_as_int(q5.w);
}

// raytrace.wgsl:253-259
PT_DEV float rand1(uint32_t &seed)
{
    seed = seed * 747796405u + 2891336453u;
    const uint32_t s = seed;
    uint32_t r = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    r = (r >> 22u) ^ r;
    return (float)r / 4294967296.0f;     // f32(4294967295.0) == 2^32
}

// raytrace.wgsl:261-265
PT_DEV float rand_normal(uint32_t &seed)
{
    const float theta = PT_TWOPI * rand1(seed);
    // (rand() returns 0 or a value in [2^-32, 1]: log1_unit is log1 without the tests no such value needs -- the same bits for
    // every one of the 2^32 values, profiles/log_unit_proof.hip)
    const float rho = ptm::sqrt_exact(-2.0f * ptm::log1_unit(rand1(seed)));
    return rho * ptm::cos1(theta);
}

// raytrace.wgsl:267-272
PT_DEV f3 rand_direction(uint32_t &seed)
{
    const float x = rand_normal(seed);
    const float y = rand_normal(seed);
    const float z = rand_normal(seed);
    return normalize(F3(x, y, z));
}

// raytrace.wgsl:283-287
PT_DEV void rand_point_in_circle(uint32_t &seed, float &px, float &py)
{
    const float theta = PT_TWOPI * rand1(seed);
    const float rho = ptm::sqrt_exact(rand1(seed));
    float s, c;
    ptm::sincos(theta, s, c);
    px = rho * c;
    py = rho * s;
}

// raytrace.wgsl:289-313; sin/cos of the rotation are passed in (same value per ray)
PT_DEV void env_uv_from_dir(const f3 &dir, float sinr, float cosr, float &u, float &v)
{
    const f3 dr = F3(dir.x * cosr - dir.z * sinr, dir.y, dir.x * sinr + dir.z * cosr);
    const float phi = ptm::atan2_1(dr.x, dr.z);
    const float theta = ptm::asin1(clamp1(dr.y, -1.0f, 1.0f));
    u = phi * PT_INVTWOPI + 0.5f;
    v = -theta * PT_INVPI + 0.5f;
}

// textureSampleLevel(environmentTexture, linear, clamp-to-edge) raytrace.wgsl:369-371
PT_DEV f3 sample_env(const float4 *env, int W, int H, float u, float v)
{
    const float x = u * (float)W - 0.5f;
    const float y = v * (float)H - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    float x0c = fminf(fmaxf(x0f, -1.0f), (float)W);
    float y0c = fminf(fmaxf(y0f, -1.0f), (float)H);
    if (x0c != x0c) x0c = 0.0f;
    if (y0c != y0c) y0c = 0.0f;
    const int x0 = (int)x0c, y0 = (int)y0c;
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
    const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    const float4 p00 = env[(size_t)ya * W + xa];
    const float4 p10 = env[(size_t)ya * W + xb];
    const float4 p01 = env[(size_t)yb * W + xa];
    const float4 p11 = env[(size_t)yb * W + xb];
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    f3 r;
    r.x = (p00.x * wx0 + p10.x * fx) * wy0 + (p01.x * wx0 + p11.x * fx) * fy;
    r.y = (p00.y * wx0 + p10.y * fx) * wy0 + (p01.y * wx0 + p11.y * fx) * fy;
    r.z = (p00.z * wx0 + p10.z * fx) * wy0 + (p01.z * wx0 + p11.z * fx) * fy;
    return r;
}

// textureSampleLevel(environmentCDFTexture, nearest sampler, clamp-to-edge): renderer.ts:82-85
PT_DEV float4 cdf_texel(const float4 *cdf, int W, int H, float u, float v)
{
    float xf = floorf(u * (float)W), yf = floorf(v * (float)H);
    xf = fminf(fmaxf(xf, 0.0f), (float)(W - 1));
    yf = fminf(fmaxf(yf, 0.0f), (float)(H - 1));
    if (xf != xf) xf = 0.0f;
    if (yf != yf) yf = 0.0f;
    return cdf[(size_t)(int)yf * W + (int)xf];
}

// raytrace.wgsl:315-349 getEnvironmentMapUV -- DEAD CODE in the reference as shipped (its call
// site :398 is commented out); runs only under mi3pt_set_env_sampling(ctx, 1).
PT_DEV void env_uv_sampled(const float4 *cdf, int W, int H, uint32_t &seed, float &u_out, float &v_out)
{
    const float r1 = rand1(seed);
    const float r2 = rand1(seed);
    float v_min = 0.0f, v_max = 1.0f;
    for (int i = 0; i < 8; i++) {
        const float v_mid = (v_min + v_max) / 2.0f;
        const float c = fmaxf(cdf_texel(cdf, W, H, 0.5f, v_mid).x, PT_EPSILON);
        if (c < r1) v_min = v_mid; else v_max = v_mid;
    }
    const float v = (v_min + v_max) / 2.0f;
    float u_min = 0.0f, u_max = 1.0f;
    for (int i = 0; i < 8; i++) {
        const float u_mid = (u_min + u_max) / 2.0f;
        const float c = fmaxf(cdf_texel(cdf, W, H, u_mid, v).y, PT_EPSILON);
        if (c < r2) u_min = u_mid; else u_max = u_mid;
    }
    u_out = (u_min + u_max) / 2.0f;
    v_out = v;
}

struct CameraFrame {      // loop-invariant part of cameraToRay, raytrace.wgsl:217-236
    float t, r;
    f3 w, u_dir, v_dir;
};

PT_DEV CameraFrame camera_frame(const RtUniforms &un)
{
    CameraFrame cf;
    const float rad = un.fov * 3.14159265358979323846f / 180.0f;    // degToRad, :213-215
    cf.t = ptm::tan1(rad / 2.0f);
    cf.r = un.aspect * cf.t;
    cf.w = normalize(neg(F3(un.cam_dir[0], un.cam_dir[1], un.cam_dir[2])));
    f3 up = F3(0.0f, 1.0f, 0.0f);
    if (fabsf(dot(cf.w, up)) > 0.99999f) up = F3(0.0f, 0.0f, 1.0f);
    cf.u_dir = normalize(cross(up, cf.w));
    cf.v_dir = cross(cf.w, cf.u_dir);
    return cf;
}

// raytrace.wgsl:219-238
PT_DEV f3 camera_direction(const CameraFrame &cf, float aspect, float uvx, float uvy)
{
    const float b = -cf.t;
    const float l = -cf.r;
    const float u = l + (cf.r - l) * uvx;
    const float v = b + (cf.t - b) * uvy;
    return normalize((cf.u_dir * u + cf.v_dir * v) - cf.w * aspect);
}

// Division of a job ticket (< 2^31) by a launch-invariant divisor d >= 1 as a multiplication: with s = ceil(log2 d)
// and m = floor(2^(31+s) / d) + 1 (which fits 32 bits), floor(t m / 2^(31+s)) = floor(t / d) for every 0 <= t < 2^31
// (Granlund & Montgomery: m d exceeds 2^(31+s) by at most d <= 2^s, so t m / 2^(31+s) overshoots t / d by less than
// 2^-s <= 1 / d).  The job decode divides three times per job; in scalar registers this is two multiplies and a shift.
struct FastDiv { uint32_t m, sh; };
PT_DEV FastDiv fast_div_of(uint32_t d)
{
    FastDiv f;
    const uint32_t s = d <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(d - 1u);
    f.sh = 31u + s;
    f.m = (uint32_t)((1ull << f.sh) / (d ? d : 1u)) + 1u;
    return f;
}
PT_DEV int fast_div(int t, const FastDiv &f) { return (int)(uint32_t)(((uint64_t)(uint32_t)t * f.m) >> f.sh); }

PT_DEV int local_to_global_row(int ly, const Tile &t)
{
    // The tile split (pt_kernels.h, Tile): local block b of rank r is block b * nranks + pos of the image, pos = r in even rounds of the
    // deal and nranks - 1 - r in odd ones -- back and forth, so that a cost that rises or falls down the image (floor, model, sky)
    // is shared out evenly (dealt one way only, rank 0 of eight got 5 % more work than rank 6: profiles/r04_rejected_and_adopted.log)
    if (t.nranks == 1) return ly;                     // (wave-uniform fast paths: no division for the whole image ...
    if (t.block_rows == 8) {                          // ... nor for the usual 8-row blocks)
        const int b = ly >> 3;
        return (((b * t.nranks + ((b & 1) ? t.nranks - 1 - t.rank : t.rank)) << 3) | (ly & 7));
    }
    const int b = ly / t.block_rows;
    return (b * t.nranks + ((b & 1) ? t.nranks - 1 - t.rank : t.rank)) * t.block_rows + (ly - b * t.block_rows);
}

PT_DEV uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One (pixel, frame) job: computeMain after its bounds check, raytrace.wgsl:429-455.
template <int VARIANT>
PT_DEV f3 shade_pixel(const RtLaunch &L, uint32_t gx, uint32_t gy, uint32_t *stack, Counters &cnt)
{
    const SceneRefs &sc = L.scene;
    const RtUniforms &un = L.un;
    cnt.pixels++;
    const float uvx = (float)gx / un.res_x;            // getUv, :247-250
    const float uvy = (float)gy / un.res_y;
    const uint32_t index = gx + gy * (uint32_t)un.res_x;
    uint32_t seed = index + un.frame * 719393u + PT_SEED;

    const CameraFrame cf = camera_frame(un);
    const f3 cam_pos = F3(un.cam_pos[0], un.cam_pos[1], un.cam_pos[2]);
    const f3 cam_dir0 = camera_direction(cf, un.aspect, uvx, uvy);
    float sinr, cosr;
    ptm::sincos(un.env_rotation, sinr, cosr);

    f3 incoming = F3(0.0f, 0.0f, 0.0f);
    for (int s = 0; s < un.samples_per_frame; s++) {
        // depth of field + anti-aliasing, :445-449
        float jx, jy, kx, ky;
        rand_point_in_circle(seed, jx, jy);
        const f3 jitter = F3(jx * (1.0f / un.res_x), jy * (1.0f / un.res_y), 0.0f);
        rand_point_in_circle(seed, kx, ky);
        const f3 jitter2 = F3(kx * un.aperture, ky * un.aperture, 0.0f);
        const f3 focal = (cam_pos + cam_dir0 * un.focal_distance) + jitter;
        f3 o = cam_pos + jitter2;
        f3 d = normalize(focal - o);

        // trace(), :373-411
        f3 light = F3(0.0f, 0.0f, 0.0f);
        f3 ray_color = F3(1.0f, 1.0f, 1.0f);
        for (int bounce = 0; bounce < un.max_bounces; bounce++) {
            Best best;
            traverse<VARIANT>(sc, o, d, stack, best, cnt);
            if (best.tri >= 0) {
                cnt.hit++;
                f3 position, normal;
                int32_t mi;
                finish_hit(sc, o, d, best, position, normal, mi);
                const float4 m0 = sc.mats[(size_t)mi * 4 + 0];   // color.rgb, -
                const float4 m1 = sc.mats[(size_t)mi * 4 + 1];   // specularColor.rgb, roughness
                const float4 m2 = sc.mats[(size_t)mi * 4 + 2];   // metalness
                const float4 m3 = sc.mats[(size_t)mi * 4 + 3];   // emissionColor.rgb, emissionStrength
                const f3 diffuse_dir = normalize(normal + rand_direction(seed));   // :279-281
                const f3 specular_dir = reflect(d, normal);
                float is_specular = 0.0f;
                if (m2.x >= rand1(seed)) is_specular = 1.0f;
                o = position;
                d = mix(diffuse_dir, specular_dir, is_specular * (1.0f - m1.w));
                const f3 emitted = xyz(m3) * m3.w;
                light = light + emitted * ray_color;
                ray_color = ray_color * mix(xyz(m0), xyz(m1), is_specular);
            } else {
                cnt.miss++;
                float u, v;
                env_uv_from_dir(d, sinr, cosr, u, v);
                if (sc.env_sampling) env_uv_sampled(sc.cdf, sc.env_w, sc.env_h, seed, u, v);        // :398
                const f3 env = sample_env(sc.env, sc.env_w, sc.env_h, u, v);
                light = light + (ray_color * env) * un.env_intensity;
                if (sc.env_sampling) {                                                                // :402-404
                    const float pdf = fmaxf(cdf_texel(sc.cdf, sc.env_w, sc.env_h, u, v).z, PT_EPSILON);
                    light = F3(light.x / pdf, light.y / pdf, light.z / pdf);
                }
                break;
            }
        }
        incoming = incoming + light;
    }
    const float n = (float)un.samples_per_frame;
    return F3(incoming.x / n, incoming.y / n, incoming.z / n);
}

PT_DEV float store_round(float v, int store_f16) { return store_f16 ? ptm::round_f16(v) : v; }

// accumulate.wgsl:18-28 for one texel
PT_DEV f3 accumulate_texel(const AccUniforms &acc, f3 color, f3 prev)
{
    float weight = 1.0f;
    if (acc.frame > 0u) weight = 1.0f / (float)acc.frame;
    weight = (acc.enabled == 1u) ? weight : 1.0f;
    return mix(prev, color, weight);
}

template <bool FUSE, int VARIANT>
__global__ void __launch_bounds__(64) k_raytrace(const RtLaunch L)
{
    __shared__ uint32_t stack_lds[PT_MAX_STACK * 64];
    const int lane = threadIdx.x;
    const int tiles_x = (L.tile.tex_w + 7) >> 3;
    const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const int gx = tile_x * 8 + (lane & 7);
    const int ly = tile_y * 8 + (lane >> 3);
    const int gy = local_to_global_row(ly, L.tile);

    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0 };
    // raytrace.wgsl:425-427
    const bool in_tex = gx < L.tile.tex_w && ly < L.tile.local_rows && gy < L.tile.tex_h;
    const bool active = in_tex && (uint32_t)gx < (uint32_t)L.un.res_x && (uint32_t)gy < (uint32_t)L.un.res_y;
    if (active) {
        f3 color = shade_pixel<VARIANT>(L, (uint32_t)gx, (uint32_t)gy, stack_lds + lane, cnt);
        const size_t idx = (size_t)ly * L.tile.tex_w + gx;
        color.x = store_round(color.x, L.store_f16);
        color.y = store_round(color.y, L.store_f16);
        color.z = store_round(color.z, L.store_f16);
        if (FUSE) {
            // accumulate.wgsl:14-16 bounds are those of the accumulate uniforms
            if ((uint32_t)gx < L.acc.res_w && (uint32_t)gy < L.acc.res_h) {
                const float4 prev = L.accum[idx];
                const f3 nc = accumulate_texel(L.acc, color, xyz(prev));
                L.accum[idx] = make_float4(store_round(nc.x, L.store_f16), store_round(nc.y, L.store_f16),
                                           store_round(nc.z, L.store_f16), 1.0f);
            }
        } else {
            L.radiance[idx] = make_float4(color.x, color.y, color.z, 1.0f);
        }
    }
    // per-block counter slots: this block is the only writer of its slot and passes
    // are stream ordered, so a plain read-modify-write by one lane is enough.
    const uint32_t s_rays = wave_sum(cnt.rays), s_box = wave_sum(cnt.box), s_tri = wave_sum(cnt.tri);
    const uint32_t s_hit = wave_sum(cnt.hit), s_miss = wave_sum(cnt.miss);
    const uint32_t s_ovf = wave_sum(cnt.overflow), s_pix = wave_sum(cnt.pixels);
    if (lane == 0 && L.block_counters) {
        uint64_t *c = L.block_counters + (size_t)blockIdx.x * CNT_COUNT;
        c[CNT_RAYS] += s_rays; c[CNT_BOX] += s_box; c[CNT_TRI] += s_tri; c[CNT_HIT] += s_hit;
        c[CNT_MISS] += s_miss; c[CNT_OVERFLOW] += s_ovf; c[CNT_PIXELS] += s_pix;
    }
}

// ---------------------------------------------------------------------------------
// VARIANT 3: persistent waves with lane refill.
//
// The per-pixel kernels above lose most of their lanes to path-length divergence: a
// wave runs until its longest path ends (up to maxBounces segments) while the average
// path is ~2 segments long.  Here a wave is a pool of 64 path slots.  Every iteration
// (1) dead slots are refilled with new (pixel, frame) jobs -- __ballot gives the dead
// mask, mbcnt ranks the dead lanes, and the ranks index consecutive pixels of the
// wave's current 8x8 tile; tiles come from one global counter, fetched one ahead --
// (2) all live lanes trace one path segment, (3) hits scatter, misses and exhausted
// paths finish and write their pixel.  Legal because every job reseeds from its own
// pixel index and frame (raytrace.wgsl:435-436): any job -> lane assignment gives the
// same pixels, bit for bit.
// ---------------------------------------------------------------------------------
PT_DEV int lane_rank(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

#ifdef MI3PT_EXPERIMENTS
template <bool FUSE>
PT_DEV void write_pixel(const RtLaunch &L, uint32_t gx, uint32_t gy, uint32_t ly, f3 color, uint32_t slot = 0u)
{
    const size_t idx = (size_t)ly * L.tile.tex_w + gx;
    color.x = store_round(color.x, L.store_f16);
    color.y = store_round(color.y, L.store_f16);
    color.z = store_round(color.z, L.store_f16);
    if (FUSE) {
        if (gx < L.acc.res_w && gy < L.acc.res_h) {
            const float4 prev = L.accum[idx];
            const f3 nc = accumulate_texel(L.acc, color, xyz(prev));
            L.accum[idx] = make_float4(store_round(nc.x, L.store_f16), store_round(nc.y, L.store_f16),
                                       store_round(nc.z, L.store_f16), 1.0f);
        }
    } else {
        L.radiance[(size_t)slot * L.slot_pixels + idx] = make_float4(color.x, color.y, color.z, 1.0f);
    }
}
#endif

// unfused store of one finished (pixel, frame slot): texel index within the slot's image
// (the storage format -- fp32, or the reference's rgba16float, renderer.ts:102 -- is a wave-uniform choice: one scalar
// branch per store, in the tuned instantiation too)
PT_DEV void write_radiance(const RtLaunch &L, uint32_t texel, uint32_t slot, f3 color)
{
    const int f16 = L.store_f16;
    L.radiance[(size_t)slot * L.slot_pixels + texel] =
        make_float4(store_round(color.x, f16), store_round(color.y, f16), store_round(color.z, f16), 1.0f);
}

#ifdef MI3PT_EXPERIMENTS
struct PathSlot {
    f3 o, d, ray_color, light, incoming;
    uint32_t gx, gy, ly, seed;
    int32_t bounce, sample;
    bool alive;
};

template <bool FUSE>
__global__ void __launch_bounds__(64) k_raytrace_persistent(const RtLaunch L)
{
    __shared__ uint32_t stack_lds[PT_MAX_STACK * 64];
    const int lane = threadIdx.x;
    uint32_t *stack = stack_lds + lane;
    const SceneRefs &sc = L.scene;
    const RtUniforms &un = L.un;
    const int tiles_x = (L.tile.tex_w + 7) >> 3;
    const int ntiles = tiles_x * ((L.tile.local_rows + 7) >> 3);
    const uint32_t res_w = (uint32_t)un.res_x, res_h = (uint32_t)un.res_y;

    const CameraFrame cf = camera_frame(un);
    const f3 cam_pos = F3(un.cam_pos[0], un.cam_pos[1], un.cam_pos[2]);
    float sinr, cosr;
    ptm::sincos(un.env_rotation, sinr, cosr);

    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0 };
    PathSlot p;
    p.alive = false;
    p.gx = p.gy = p.ly = p.seed = 0u;
    p.bounce = p.sample = 0;
    p.o = p.d = p.ray_color = p.light = p.incoming = F3(0.0f, 0.0f, 0.0f);

    // Starts the slot's next camera path, or finishes the pixel when all samples are done
    // (raytrace.wgsl:441-455).  A zero-length path (maxBounces <= 0) contributes nothing.
    auto next_path = [&]() {
        for (;;) {
            if (p.sample >= un.samples_per_frame) {
                const float n = (float)un.samples_per_frame;
                write_pixel<FUSE>(L, p.gx, p.gy, p.ly, F3(p.incoming.x / n, p.incoming.y / n, p.incoming.z / n));
                p.alive = false;
                return;
            }
            const float uvx = (float)p.gx / un.res_x, uvy = (float)p.gy / un.res_y;
            const f3 dir0 = camera_direction(cf, un.aspect, uvx, uvy);
            float jx, jy, kx, ky;
            rand_point_in_circle(p.seed, jx, jy);
            const f3 jitter = F3(jx * (1.0f / un.res_x), jy * (1.0f / un.res_y), 0.0f);
            rand_point_in_circle(p.seed, kx, ky);
            const f3 jitter2 = F3(kx * un.aperture, ky * un.aperture, 0.0f);
            const f3 focal = (cam_pos + dir0 * un.focal_distance) + jitter;
            p.o = cam_pos + jitter2;
            p.d = normalize(focal - p.o);
            p.bounce = 0;
            p.light = F3(0.0f, 0.0f, 0.0f);
            p.ray_color = F3(1.0f, 1.0f, 1.0f);
            if (un.max_bounces > 0) {
                p.alive = true;
                return;
            }
            p.incoming = p.incoming + p.light;
            p.sample++;
        }
    };

    auto fetch_tile = [&]() -> int {
        int t = 0;
        if (lane == 0) t = (int)atomicAdd(L.tile_counter, 1u);
        return __builtin_amdgcn_readfirstlane(t);
    };

    int cur_tile = 0, cur_used = 64;
    int next_tile = fetch_tile();
    bool feed_empty = false;

    for (;;) {
        // ---- (1) refill dead slots from the wave's tile
        unsigned long long dead = __ballot(!p.alive);
        while (dead != 0ull && !feed_empty) {
            if (cur_used >= 64) {
                cur_tile = next_tile;
                if (cur_tile >= ntiles) { feed_empty = true; break; }
                next_tile = fetch_tile();
                cur_used = 0;
            }
            const int take = min((int)__popcll(dead), 64 - cur_used);
            const int rank = lane_rank(dead);
            if (!p.alive && rank < take) {
                const int j = cur_used + rank;
                const int px = (cur_tile % tiles_x) * 8 + (j & 7);
                const int ply = (cur_tile / tiles_x) * 8 + (j >> 3);
                const int pgy = local_to_global_row(ply, L.tile);
                // raytrace.wgsl:425-427
                const bool ok = px < L.tile.tex_w && ply < L.tile.local_rows && pgy < L.tile.tex_h &&
                                (uint32_t)px < res_w && (uint32_t)pgy < res_h;
                if (ok) {
                    p.gx = (uint32_t)px; p.gy = (uint32_t)pgy; p.ly = (uint32_t)ply;
                    cnt.pixels++;
                    p.seed = (p.gx + p.gy * res_w) + un.frame * 719393u + PT_SEED;
                    p.sample = 0;
                    p.incoming = F3(0.0f, 0.0f, 0.0f);
                    next_path();
                }
            }
            cur_used += take;
            dead = __ballot(!p.alive);
        }
        if (__ballot(p.alive) == 0ull) break;

        // ---- (2) one path segment for every live slot, (3) shade
        if (p.alive) {
            Best best;
            traverse_packets(sc, p.o, p.d, stack, best, cnt);
            bool ended;
            if (best.tri >= 0) {
                cnt.hit++;
                f3 position, normal;
                int32_t mi;
                finish_hit(sc, p.o, p.d, best, position, normal, mi);
                const float4 m0 = sc.mats[(size_t)mi * 4 + 0];
                const float4 m1 = sc.mats[(size_t)mi * 4 + 1];
                const float4 m2 = sc.mats[(size_t)mi * 4 + 2];
                const float4 m3 = sc.mats[(size_t)mi * 4 + 3];
                const f3 diffuse_dir = normalize(normal + rand_direction(p.seed));
                const f3 specular_dir = reflect(p.d, normal);
                float is_specular = 0.0f;
                if (m2.x >= rand1(p.seed)) is_specular = 1.0f;
                p.o = position;
                p.d = mix(diffuse_dir, specular_dir, is_specular * (1.0f - m1.w));
                const f3 emitted = xyz(m3) * m3.w;
                p.light = p.light + emitted * p.ray_color;
                p.ray_color = p.ray_color * mix(xyz(m0), xyz(m1), is_specular);
                p.bounce++;
                ended = p.bounce >= un.max_bounces;
            } else {
                cnt.miss++;
                float u, v;
                env_uv_from_dir(p.d, sinr, cosr, u, v);
                const f3 env = sample_env(sc.env, sc.env_w, sc.env_h, u, v);
                p.light = p.light + (p.ray_color * env) * un.env_intensity;
                ended = true;
            }
            if (ended) {
                p.incoming = p.incoming + p.light;
                p.sample++;
                next_path();
            }
        }
    }

    // Self-cleaning work queue: the last wave to leave resets the head and the exit counter, so
    // launches need no memset in front of them (a memset kernel would have to wait for a free
    // slot among the previous batch's persistent waves).
    if (lane == 0) {
        const uint32_t left = atomicAdd(L.tile_counter + 1, 1u);
        if (left == gridDim.x - 1u) {
            __hip_atomic_store(L.tile_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(L.tile_counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const uint32_t s_rays = wave_sum(cnt.rays), s_box = wave_sum(cnt.box), s_tri = wave_sum(cnt.tri);
    const uint32_t s_hit = wave_sum(cnt.hit), s_miss = wave_sum(cnt.miss);
    const uint32_t s_ovf = wave_sum(cnt.overflow), s_pix = wave_sum(cnt.pixels);
    if (lane == 0 && L.block_counters) {
        uint64_t *c = L.block_counters + (size_t)blockIdx.x * CNT_COUNT;
        c[CNT_RAYS] += s_rays; c[CNT_BOX] += s_box; c[CNT_TRI] += s_tri; c[CNT_HIT] += s_hit;
        c[CNT_MISS] += s_miss; c[CNT_OVERFLOW] += s_ovf; c[CNT_PIXELS] += s_pix;
    }
}

#endif      // MI3PT_EXPERIMENTS (variant 3)

// ---------------------------------------------------------------------------------
// VARIANT 4: persistent waves, per-lane state machine.
//
// Variant 3 still runs every segment's walk to the longest lane's length.  Here each
// lane is in one of three modes and the wave alternates between two kinds of step:
//   TRAVERSE  the lane is inside rayBVHIntersect: stack pointer, best hit and stack
//             (LDS) persist across steps; a "walk step" pops ONE stack entry per lane.
//   SHADE     the walk is over (stack empty, root missed or 64-entry abort): the lane
//             waits for the next "service step", which shades hits / looks up the
//             environment, finishes pixels, refills dead lanes with new jobs and starts
//             the next segments (root box test included).
//   DEAD      no job.
// Walk steps run while at least walk_min lanes are walking (or nothing else can make
// progress); otherwise a service step runs with every waiting lane at once.  Each lane
// still executes exactly the reference's sequence of tests for its own ray, so results
// and counters are unchanged.
// ---------------------------------------------------------------------------------
enum { M_DEAD = 0, M_TRAV = 1, M_SHADE = 2, M_PATH = 3 };

// Per-segment constants of the CULL walk's distance bound.  A triangle with
// E = |e1| |e2| and |e1| + |e2| <= L_max that the reference's Moller-Trumbore code accepts with
// t <= best.t lies within
//     delta = W (u / EPSILON) (best.t |d|^2 + 1.65 L_max |d|),   W = E c1(E),  u = 2^-24
// of the point o + t d, hence so does every box that holds it (context: prepare_cull computes W
// per child and folds the rest into SceneRefs::cull_ka / cull_kb).  Per segment:
//     delta <= W * (Ka * best.t + Kb),   Ka = cull_ka |d|^2,   Kb = cull_kb max(1, |d|^2) >= cull_kb |d|.
// Rays the analysis does not cover (a plain-division ray, |d| > 2, non-finite) get Ka = Kb =
// +infinity: nothing is skipped (a child that must never be skipped carries W = +infinity).
PT_DEV void cull_setup(const f3 &d, const RayPre &pre, float scene_ka, float scene_kb, float &ka, float &kb)
{
    const float dd = dot(d, d) * 1.0000005f;            // >= |d|^2 (three products, two sums: 3 u relative)
    const bool ok = (pre.flags & 8u) == 0u && dd <= 4.0f;
    ka = ok ? scene_ka * dd : __builtin_inff();
    kb = ok ? scene_kb * fmaxf(1.0f, dd) : __builtin_inff();
}

// ---------------------------------------------------------------------------------
// The launch-invariant scalars of the state-machine kernel's SERVICE step: the camera frame, sin / cos of the environment
// rotation, the root box, 1 / resolution, the job-decode divisors, and the launch block itself (pointers, uniforms, tile).
// None of it is touched by node or triangle steps, yet held in scalar registers for the whole kernel it was ~70 SGPRs
// too many: the compiler spilled them to vector-register lanes and reloaded them with v_readlane, 143 of them in the
// service step (round 2: `SGPRs Spill: 67`).  The tuned instantiations therefore keep it in memory: k_rt_service_setup
// computes the block once per launch -- with the very device functions the kernel used to call in its prologue, so the
// bits are the same -- and every service step loads what it uses with scalar loads (load_const_block: s_load through a
// constant-address-space pointer whose offset the compiler cannot see through, so the loads are not hoisted back out of
// the loop).  The diagnostic twins compute the same block in their prologue and keep it in registers, as before.
// ---------------------------------------------------------------------------------
struct RtService {
    RtLaunch L;
    CameraFrame cf;
    f3 cam_pos;
    float sinr, cosr;
    float root_mn[3], root_mx[3];
    float inv_res_x, inv_res_y, spf_f;
    uint32_t res_w, res_h;
    int32_t pinhole, res_ordinary;
    int32_t tiles_x, ntiles_frame, ntiles;
    int32_t grp_jobs, grp_full, grp_last;
    FastDiv dv_frame, dv_grp, dv_gs, dv_last, dv_tx, dv_w;
};
static_assert(sizeof(RtService) % 4 == 0, "loaded as dwords");
size_t service_block_bytes() { return sizeof(RtService); }

// What the state-machine kernel's prologue computes from the launch block (per lane, identically in every lane).
PT_DEV RtService compute_service(const RtLaunch &L, bool scene_has_nodes)
{
    RtService S;
    const RtUniforms &un = L.un;
    S.L = L;
    S.cf = camera_frame(un);
    S.cam_pos = F3(un.cam_pos[0], un.cam_pos[1], un.cam_pos[2]);
    ptm::sincos(un.env_rotation, S.sinr, S.cosr);
    float4 root0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), root1 = root0;
    if (scene_has_nodes) { root0 = L.scene.nodes[0]; root1 = L.scene.nodes[1]; }
    S.root_mn[0] = root0.x; S.root_mn[1] = root0.y; S.root_mn[2] = root0.z;
    S.root_mx[0] = root1.x; S.root_mx[1] = root1.y; S.root_mx[2] = root1.z;
    // per-pixel divisions by launch-invariant divisors: 1 / resolution once, and pixel / resolution
    // as an exact quotient from that reciprocal (ptm::div_pre; pixel indices are 0 or >= 1, so the
    // only proviso is a resolution of ordinary magnitude -- otherwise the plain division runs)
    S.inv_res_x = 1.0f / un.res_x;
    S.inv_res_y = 1.0f / un.res_y;
    S.res_ordinary = (un.res_x >= 9.5367431640625e-07f && un.res_x <= 1.099511627776e12f &&
                      un.res_y >= 9.5367431640625e-07f && un.res_y <= 1.099511627776e12f) ? 1 : 0;
    S.spf_f = (float)un.samples_per_frame;
    S.res_w = (uint32_t)un.res_x;
    S.res_h = (uint32_t)un.res_y;
    // thin lens off (aperture exactly 0, no -0 camera coordinate): see the camera path start
    S.pinhole = (un.aperture == 0.0f && __float_as_uint(un.cam_pos[0]) != 0x80000000u &&
                 __float_as_uint(un.cam_pos[1]) != 0x80000000u && __float_as_uint(un.cam_pos[2]) != 0x80000000u) ? 1 : 0;
    S.tiles_x = (L.tile.tex_w + 7) >> 3;
    // A launch covers L.nframes consecutive frames: job = (frame slot, 8x8 tile)
    S.ntiles_frame = S.tiles_x * ((L.tile.local_rows + 7) >> 3);
    S.ntiles = S.ntiles_frame * L.nframes;
    S.grp_jobs = (L.job_group > 0 && L.job_group < S.ntiles_frame) ? L.job_group * L.nframes : 0;
    S.grp_full = S.grp_jobs ? S.ntiles_frame / L.job_group : 0;
    S.grp_last = S.grp_jobs ? S.ntiles_frame - S.grp_full * L.job_group : 1;       // tiles in the (shorter) last group
    S.dv_frame = fast_div_of((uint32_t)S.ntiles_frame);
    S.dv_grp = fast_div_of((uint32_t)(S.grp_jobs ? S.grp_jobs : 1));
    S.dv_gs = fast_div_of((uint32_t)(L.job_group > 0 ? L.job_group : 1));
    S.dv_last = fast_div_of((uint32_t)(S.grp_last > 0 ? S.grp_last : 1));
    S.dv_tx = fast_div_of((uint32_t)S.tiles_x);
    S.dv_w = fast_div_of((uint32_t)(L.tile.tex_w > 0 ? L.tile.tex_w : 1));
    return S;
}

// The pixel-only part of a camera ray (raytrace.wgsl:219-238 and the focal point's base, :446): uv = pixel / resolution,
// dir0 = cameraToRay's direction, base = cam_pos + dir0 * focalDistance.  A batched launch forms it for every FRAME of a pixel
// (256 times per launch, ~65 of a camera-path start's ~180 vector instructions, at ~30 of 64 lanes); this kernel forms it once
// per camera and image size with the very same device functions, full waves, and the service step loads it (one 16-byte load,
// coalesced along the 8-pixel rows of a tile, issued ahead of the disk sample that hides it).
__global__ void __launch_bounds__(256) k_camera_base(const RtLaunch L, float4 *out)
{
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ly = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= L.tile.tex_w || ly >= L.tile.local_rows) return;
    const RtUniforms &un = L.un;
    const CameraFrame cf = camera_frame(un);
    const int pgy = local_to_global_row(ly, L.tile);
    const bool ordinary = un.res_x >= 9.5367431640625e-07f && un.res_x <= 1.099511627776e12f &&
                          un.res_y >= 9.5367431640625e-07f && un.res_y <= 1.099511627776e12f;      // (compute_service: res_ordinary)
    float uvx, uvy;
    if (ordinary) { uvx = div_pre((float)px, un.res_x, 1.0f / un.res_x); uvy = div_pre((float)pgy, un.res_y, 1.0f / un.res_y); }
    else { uvx = (float)px / un.res_x; uvy = (float)pgy / un.res_y; }
    const f3 dir0 = camera_direction(cf, un.aspect, uvx, uvy);
    const f3 base = F3(un.cam_pos[0], un.cam_pos[1], un.cam_pos[2]) + dir0 * un.focal_distance;
    out[(size_t)ly * L.tile.tex_w + px] = make_float4(base.x, base.y, base.z, 0.0f);
}

void launch_camera_base(const RtLaunch &L, float4 *out, hipStream_t s)
{
    if (L.tile.tex_w <= 0 || L.tile.local_rows <= 0) return;
    hipLaunchKernelGGL(k_camera_base, dim3((L.tile.tex_w + 63) / 64, (L.tile.local_rows + 3) / 4), dim3(256), 0, s, L, out);
}

__global__ void __launch_bounds__(64) k_rt_service_setup(const RtLaunch L, RtService *out)
{
    const RtService S = compute_service(L, L.scene.nnodes != 0);
    if (threadIdx.x == 0) *out = S;
}

// A block of launch-invariant scalars read back from memory with scalar loads.  The opaque offset (always 0) keeps the
// compiler from hoisting the loads out of the loop they are issued in; the constant address space makes them s_load.
template <class T>
PT_DEV T load_const_block(const T *g)
{
    static_assert(sizeof(T) % 4 == 0 && __is_trivially_copyable(T), "dwords");
    uint32_t off = 0;
    asm volatile("" : "+s"(off));
    typedef const __attribute__((address_space(4))) uint32_t *cptr;
    const cptr p = (cptr)((uintptr_t)g + off);
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; i++) w[i] = p[i];
    T t;
    __builtin_memcpy(&t, w, sizeof(T));
    return t;
}
// The same values made wave-uniform by readfirstlane (a block computed by vector instructions: the diagnostic twins)
template <class T>
PT_DEV T uniform_block(const T &v)
{
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; i++) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[i]);
    T t;
    __builtin_memcpy(&t, w, sizeof(T));
    return t;
}

#ifndef PT_CW_NO_PAIR_TEST
#define PT_CW_NO_PAIR_TEST 0
#endif
#define PT_SM_LDS_DEPTH pt::SM_LDS_DEPTH
#ifndef PT_LONG_QUEUE
#define PT_LONG_QUEUE 16
#endif
#ifndef PT_DEEP_LEAF_MIN
#define PT_DEEP_LEAF_MIN 32
#endif
#ifndef PT_DEEP_LEAF_CAP
#define PT_DEEP_LEAF_CAP 10
#endif
#ifdef PT_X_TOP_CW
#define PT_X_TOP_CW_N PT_X_TOP_CW
#else
#define PT_X_TOP_CW_N 8
#endif
#define PT_SM_TOP_PACKETS 32       // node packets staged in LDS per wave (2 KB: 16 waves per CU still fit): the top 5 levels
// TOPLDS = true additionally stages the first PT_SM_TOP_PACKETS node packets in LDS (kernel
// variant 6).  Measured on MI355X (round 1): no gain -- the kernel is
// VALU-issue-bound and the top of the tree is L1-resident anyway, while the second load path
// costs registers -- so the shipped default keeps TOPLDS = false.
// (5 waves per SIMD = 96 VGPRs spills 53 registers and runs 28 % slower: profiles/r01_g_rejected_experiments.log)
//
// CULL (kernel variant 9, needs DEFER): exact-image distance culling.  The reference walk has no
// upper bound by the current hit (raytrace.wgsl:118-152, 154-203): it tests every box the ray
// touches.  A child whose box the ray enters at tmin is skipped here when, in essence,
//     tmin  -  W * (Ka * best.t + Kb) / |d_k|  >  best.t
// (precisely: tnear_i - W (Ka best.t + Kb) / |d_i| > best.t (1 + 2^-20) on ANY axis i skips the child)
// where W (packet, 16 bits per child, rounded up) bounds |e1| * |e2| * c1 over the triangles below
// the child and Ka, Kb are per-segment constants of the ray (cull_setup).  DESIGN.md section 3a proves
// that every triangle below such a child, had it been tested, would have been rejected or have
// returned t > best.t in the reference's own fp32 Moller-Trumbore arithmetic -- so the closest
// hit, its (t, u, v) and the tie rule are untouched and images stay bit-identical, while the
// box / triangle COUNTERS drop below the reference's.  Children are pushed far first, near last.
//
// WIDE (kernel variant 10, needs CULL): the walk runs on 4-ary "wide packets" (pt_kernels.h) -- up
// to four child boxes per node step, half as many dependent round trips per ray.  The leaves reached
// are exactly the reference's (monotone slab test under nesting).
//
// FILT (WIDE only): the filtered slab test -- approximate quotients decide, the exact test runs for a box whose interval
// ends lie within 2^-21 of each other (slab_q0).  The context chooses it per scene (SceneRefs::slab_filter): boxes that
// are flat on an axis and hit through that face (the two triangles of a floor, axis-aligned quads) ALWAYS land in the
// exact test, so a scene whose walks are short and meet such leaves often is faster without the filter.
// YMAX (FILT only): the culling condition (S) of DESIGN.md 3a evaluated on the axis that sets the box's entry distance
// only, with the largest of the three |RN(1/d_i)| -- one operation per child instead of four; it skips less, so the
// context chooses it only for scenes whose culling margins are negligible (SceneRefs::cull_ymax).
//
// DIAG = false: the lean build every ordinary launch runs (below).  TOPLDS, LITE: experiment builds only.  LITE (with DIAG = false):
// the lean build plus the wave-uniform lane counts per kind of step (scalar registers) and two clock reads, written once at exit
// -- how many lanes each kind of step serves in the kernel that ships, not in its four-wave diagnostic twin (round-3 verdict).
// CW (with WIDE): the walk on compressed wide packets and 64-byte triangle records (variant 13; cwide_hit above).
// WMIN: the lean build's walk_min (a constant there: as a launch parameter in a scalar register it cost the 870 k-triangle scene 0.8 %
// and the demo scene 1.5 %, profiles/r04_o_walkmin2.log): 32, or 44 for the deep walks of very large trees (compressed packets only).
// WAVES: resident waves per SIMD of a lean build (5: 96 registers, SM_LDS_DEPTH stack entries in LDS; 6: 80 registers, SM_LDS_DEPTH_SIX).
// W8 (with CW): the walk on EIGHT-wide compressed packets (variant 14; pt_kernels.h CW8Packet): a node step tests eight boxes and leaves two
// 8-bit hit masks -- the internal children in the visiting order `slot ^ octant`, the leaves by slot -- pushes at most one 64-bit node entry
// and one 32-bit leaf entry, and the next step pops the nearest internal child by a find-first-bit: no sort, no per-child references.
template <bool DEFER, bool CULL, bool WIDE, bool FILT, bool YMAX, bool DIAG, bool TOPLDS = false, bool LITE = false, bool CW = false,
          int WMIN = PT_DEFAULT_WALK_MIN, int WAVES = SM_TUNED_WAVES_PER_SIMD, bool W8 = false>
__global__ void __launch_bounds__(64, DIAG ? SM_OTHER_WAVES_PER_SIMD : WAVES) k_raytrace_sm(const RtLaunch L)
{
    static_assert(!CULL || DEFER, "the culling walks park their leaves");
    static_assert(!W8 || (CW && !DIAG && !TOPLDS), "the 8-wide walk runs on compressed packets, lean builds only");
    static_assert((!WIDE || CULL) && (!FILT || WIDE) && (!YMAX || FILT) && (!CW || FILT), "WIDE needs CULL, FILT needs WIDE, YMAX and CW need FILT");
    // DIAG = false (the shipped walks' batched launches when no diagnostic buffer is bound): the per-wave step statistics
    // and stamps are compiled out -- two dozen scalar registers that the walk loop's own scalars were spilled for
    uint64_t *const wave_times = DIAG ? L.wave_times : nullptr;
    const bool count_on = DIAG ? wave_times != nullptr : LITE;        // the step / lane counts (LITE: always; the lean build: never)
    // ... and the step-voting knobs are the defaults as constants (launch_raytrace sends any other setting to the DIAG twin)
    const int k_walk_min = DIAG ? L.walk_min : WMIN, k_leaf_min = DIAG ? L.leaf_min : (WMIN == PT_DEEP_WALK_MIN && CW && !W8 ? PT_DEEP_LEAF_MIN : PT_DEFAULT_LEAF_MIN);
    const int k_shade_split = DIAG ? L.shade_split : PT_DEFAULT_SHADE_SPLIT, k_tail_policy = DIAG ? L.tail_policy : PT_DEFAULT_TAIL_POLICY;
    const int k_job_chunk = DIAG ? L.job_chunk : PT_DEFAULT_JOB_CHUNK;
    const bool k_tri_pair = DIAG ? L.tri_pair != 0 : !PT_CW_NO_PAIR_TEST;
    constexpr bool TUNED = !DIAG;
    // ... and, in the lean builds of the culling walks, so are the conditions every scene that admits those walks meets at an
    // ordinary resolution: a scene with nodes whose root is an internal node with a guard-range box, a resolution of ordinary
    // magnitude (launch_assumptions_hold; a launch that fails them runs the lean build of variant 7 or 4, which assumes nothing)
    constexpr bool ASSUME = TUNED && CULL;
    static_assert(WAVES == SM_TUNED_WAVES_PER_SIMD || (WAVES == SM_SIX_WAVES_PER_SIMD && !DIAG), "five waves per SIMD, or six for a lean build");
    // PARKG: the six-wave build for very large trees (walk_min 44): its whole 6 400-byte LDS share is stack (25 entries: deep walks pay
    // for every entry that leaves LDS), and a path's throughput and collected light, touched by the service step only, rest in the
    // wave's slice of L.park in memory -- read where shading starts, ahead of the loads it waits for anyway, written where a segment starts
    constexpr bool SIX = WAVES == SM_SIX_WAVES_PER_SIMD && WAVES != SM_TUNED_WAVES_PER_SIMD;
#ifdef PT_X_PARKG_ALL
    constexpr bool PARKG = SIX;
#else
    constexpr bool PARKG = SIX && WMIN == PT_DEEP_WALK_MIN;
#endif
    constexpr int DEPTH = PARKG ? SM_LDS_DEPTH_SIX_DEEP : (SIX ? SM_LDS_DEPTH_SIX : PT_SM_LDS_DEPTH);      // LDS stack entries per lane
    // (the builds for very large trees: a leaf list of PT_DEEP_LEAF_CAP and a triangle step from PT_DEEP_LEAF_MIN lanes on -- round 6,
    // profiles/r06_c_leaf_step_sweep.log: the forest's walks park more leaves per ray than any other scene's)
    constexpr int LCAP = W8 ? SM_W8_LEAF_CAP : (WMIN == PT_DEEP_WALK_MIN && CW && TUNED && DEPTH >= SM_LDS_DEPTH ? PT_DEEP_LEAF_CAP : SM_CULL_LEAF_CAP), NCAP = DEPTH - LCAP;     // culling walks: leaf list / node slots in LDS
    constexpr int NCAP8 = (DEPTH - SM_W8_LEAF_CAP) / 2;             // 8-wide walk: 64-bit node entries in LDS (two dwords each, from slot 0)
    static_assert(!W8 || (NCAP8 >= SM_W8_MIN_LDS_NODES && 2 * SM_W8_OVERFLOW_NODES <= SM_OVERFLOW_ENTRIES), "8-wide walk: node entries in LDS + overflow slice");
    // The first PT_SM_LDS_DEPTH stack entries live in LDS ([depth][lane]: conflict free);
    // deeper entries (rare: the stack holds about one entry per tree level) go to this
    // wave's slice of a global overflow area, so the 64-entry abort semantics are kept
    // while a wave only needs 8 KB of LDS.
    __shared__ uint32_t stack_lds[DEPTH * 64];
    // BVH node packets staged in LDS: packets are numbered breadth-first, so the first
    // PT_SM_TOP_PACKETS of them ARE the top levels of the tree (raytrace.ts:667-678), which
    // every ray walks.  One-wave workgroups (a multi-wave workgroup would hold its LDS and
    // wave slots until its slowest wave has drained) => a private 4 KB copy per wave.
    // (compressed-wide walk, build flag PT_X_TOP_CW = n <= 8: 20 waves x 7 680 B leave 512 B per wave of the CU's 160 KiB -- eight 64-byte packets)
    constexpr int TOPN = (TOPLDS && CW) ? PT_X_TOP_CW_N : PT_SM_TOP_PACKETS;
    __shared__ float4 top_lds[TOPLDS ? TOPN * 4 : 1];
    // A path's throughput and collected light are only touched by the service step; between service steps they rest here
    // (6 floats per lane, [k][lane]) instead of in six registers carried through every node and triangle step
    __shared__ float park_lds[PARKG ? 1 : 6 * 64];
    const int lane = threadIdx.x;
    float *park = park_lds + (PARKG ? 0 : lane);
    float *const parkg = PARKG ? L.park + (size_t)blockIdx.x * (6 * 64) + lane : nullptr;
    uint32_t *stack = stack_lds + lane;
    const uint32_t top_cap = (uint32_t)L.top_packets < (uint32_t)TOPN ? (uint32_t)L.top_packets : (uint32_t)TOPN;
    const uint32_t ntop = !TOPLDS ? 0u : (CW ? top_cap : (L.scene.npackets < top_cap ? L.scene.npackets : top_cap));      // (CW: the host caps top_packets by the number of wide packets)
    if (TOPLDS) {
        for (uint32_t i = (uint32_t)lane; i < ntop * 4u; i += 64u) top_lds[i] = CW ? L.scene.cwide[i] : L.scene.packets[i];
        __syncthreads();
    }
    uint32_t *ovf = L.stack_overflow + (size_t)blockIdx.x * (SM_OVERFLOW_ENTRIES * 64) + lane;      // (a slice of SM_OVERFLOW_ENTRIES >= PT_MAX_STACK - DEPTH entries per lane)
    // LDS and the overflow slice are accessed by separate instructions (a pointer select
    // between the two address spaces would compile to flat_* accesses, which wait for every
    // outstanding memory operation): the LDS access is unconditional, the overflow access a
    // rare branch kept apart by the opaque asm.
    auto st_load = [&](int i) -> uint32_t {
        uint32_t v = stack[(i < DEPTH ? i : 0) * 64];
        asm volatile("" : "+v"(v));          // keep this a ds_read of its own
        if (i >= DEPTH) v = ovf[(i - DEPTH) * 64];
        return v;
    };
    auto st_store = [&](int i, uint32_t v) {
        if (i >= DEPTH) ovf[(i - DEPTH) * 64] = v;
        else stack[i * 64] = v;
    };
    const SceneRefs &sc = L.scene;
    (void)0;
    int sp = 0;
    int nl = 0;              // DEFER: leaves waiting in this lane's list (LDS slots DEPTH - 1, DEPTH - 2, ...)
    // Culling walks: node entries [0, NCAP) in LDS, deeper ones in the overflow slice; leaves in the LCAP slots on top.
    // An entry that is not taken (box missed / skipped) may still be written to the free slot `sp` (no select on the
    // store); in the overflow range only entries that are taken are written.
    auto cull_pop = [&]() -> uint32_t {
        sp--;
        uint32_t v = stack[(sp < NCAP ? sp : 0) * 64];
        asm volatile("" : "+v"(v));          // keep this a ds_read of its own
        if (sp >= NCAP) v = ovf[(sp - NCAP) * 64];
        return v;
    };
    auto cull_push = [&](uint32_t ref, bool take) {
        const bool lf = (ref & PT_REF_LEAF) != 0u;
        if (lf || sp < NCAP) stack[(lf ? DEPTH - 1 - nl : sp) * 64] = ref;       // (a leaf entry keeps its leaf bit: stripped where the triangle step reads it)
        else if (take) ovf[(sp - NCAP) * 64] = ref;
        nl += (take && lf) ? 1 : 0;
        sp += (take && !lf) ? 1 : 0;
    };
    // The same two without the overflow slice, for a step in which no lane of the wave can reach it (a wave-uniform
    // test at the top of the node step; almost every step): no branches, an empty entry (PT_REF_NONE, which has the
    // leaf bit) is written to the free leaf slot and not counted.
    auto flat_pop = [&]() -> uint32_t { sp--; return stack[sp * 64]; };
    auto flat_push = [&](uint32_t ref) {
        const bool lf = (ref & PT_REF_LEAF) != 0u;
        stack[(lf ? DEPTH - 1 - nl : sp) * 64] = ref;
        nl += (lf && ref != PT_REF_NONE) ? 1 : 0;
        sp += lf ? 0 : 1;
    };
    // The service step's launch-invariant scalars (RtService).  Tuned instantiation: in memory, written by
    // k_rt_service_setup before this launch; each service step loads what it uses (nothing of it lives in registers during
    // the walk).  Otherwise: computed here, made wave-uniform by readfirstlane, kept in scalar registers.
    RtService S0;
    if constexpr (!TUNED) S0 = uniform_block(compute_service(L, sc.nnodes != 0));
    // diagnostic stamps (only when a buffer is bound): wall clock (100 MHz) and shader clock
    const uint64_t t_begin_rt = (wave_times || LITE) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const uint64_t t_begin_clk = (wave_times || LITE) ? __builtin_amdgcn_s_memtime() : 0ull;
    uint64_t t_empty_rt = 0ull;
    // diagnostic step statistics (wave-uniform; stored with the stamps)
    uint32_t st_walk_steps = 0, st_walk_lanes = 0, st_leaf_lanes = 0, st_service_steps = 0;
    uint32_t st_shade_lanes = 0, st_hit_lanes = 0, st_path_lanes = 0, st_segment_lanes = 0, st_tri_steps = 0, st_parked = 0;
    uint32_t st_hit_steps = 0, st_b_steps = 0;       // service steps that served the hit group / the miss + path group
    uint32_t st_tail_node = 0, st_tail_tri = 0, st_tail_service = 0, st_tail_lanes = 0, st_live_at_empty = 0;   // after the queue ran empty
    uint64_t st_cyc_node = 0, st_cyc_tri = 0, st_cyc_service = 0, st_mark = t_begin_clk;     // shader cycles per kind of step
    int st_kind = 2;
#ifdef PT_DIAG_SERVICE      // experiment build: the service step's parts (3 hit shading, 4 miss shading, 5 refill + camera paths, 6 segment starts)
    uint64_t st_cyc_part[4] = { 0, 0, 0, 0 };
#endif
    auto st_switch = [&](int kind) {     // diagnostic only: the time since the last switch belongs to the step that ran
        const uint64_t now = __builtin_amdgcn_s_memtime();
        const uint64_t dt = now - st_mark;
        if (st_kind == 0) st_cyc_node += dt; else if (st_kind == 1) st_cyc_tri += dt;
#ifdef PT_DIAG_SERVICE
        else if (st_kind >= 3) st_cyc_part[st_kind - 3] += dt;
#endif
        else st_cyc_service += dt;
        st_mark = now;
        st_kind = kind;
    };
    const int lcap = CULL ? LCAP : L.scene.leaf_cap;      // DEFER: capacity of a lane's leaf list

    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0 };      // per-lane: only what the in-order walk counts under divergence
    uint32_t u_rays = 0, u_box = 0, u_tri = 0, u_hit = 0, u_miss = 0, u_pix = 0, u_slow = 0;     // wave-uniform (scalar) counts
    int mode = M_DEAD;
    f3 o = F3(0.0f, 0.0f, 0.0f), d = o, ray_color = o, light = o;
    // What a lane carries through its walks besides the ray: the texel index of its pixel within the rank's image (the pixel's
    // coordinates are only needed for the camera ray, formed in the service step) and frame slot << 16 | bounce.  The sum over a
    // multi-sample frame's samples and their count rest in the pixel's texel of the frame's radiance slot (see the path end).
    uint32_t gx = 0u, seed = 0u, slot = 0u;
    uint32_t lane_cost = 0u;       // DIAG, a measuring launch (L.tile_cost): what this lane's current path has cost so far
    Best best;
    best.t = PT_INF; best.u = best.v = 0.0f; best.tri = -1;
    RayPre pre;
    pre.ix = pre.iy = pre.iz = 0.0f;
    pre.flags = 8u;
    float cull_ka = __builtin_inff(), cull_kb = __builtin_inff();      // CULL: per-segment constants of the distance bound

    // Job tickets are drawn one draw ahead of their use and L.job_chunk at a time while the queue is long: one atomic
    // per job, all on one address, paces the whole chip at ~80 M jobs/s -- the rate of the 2 k-triangle scene (it ran
    // 36 % faster with 4 tickets per draw, the 870 k-triangle scene 2 %).  Near the end of the queue (fewer than two
    // draws per wave left) tickets are drawn singly, so that no wave sits on jobs while others run dry.
    int have = 0, have_at = 0;           // tickets in hand: have_at, have_at + 1, ... (`have` of them)
    auto draw = [&](const RtService &S) {
        const int ntiles = S.ntiles;
        // (round 6: twice the tickets per draw while the queue is VERY long -- more than PT_LONG_QUEUE draws of that size per wave left: the
        // head's atomic paces a 5 M-job launch measurably (+1 % with eight per draw), while a launch of a few hundred thousand jobs loses to it)
        const int left = ntiles - have_at;
        const int c = (TUNED && left > PT_LONG_QUEUE * 2 * k_job_chunk * (int)gridDim.x) ? 2 * k_job_chunk
                      : (left > 2 * k_job_chunk * (int)gridDim.x ? k_job_chunk : 1);
        int t = 0;
        if (lane == 0) {
            t = (int)atomicAdd(S.L.tile_counter, (uint32_t)c);
            // When the queue is about to run dry (half a grid of jobs left: a lead of some tens of
            // microseconds over the first exiting wave, which covers the command processor's wake-up
            // and the dispatch) this launch announces its drain.  The host holds the next launch
            // back (a stream wait on this word) until then, so that launches run back to back with
            // only their tails overlapping instead of queueing for slots behind each other.
            const int drain_mark = max(ntiles - (int)(gridDim.x >> 1), 0);
            // (an atomic max: the host may have stepped past this mark while releasing a launch that looked stalled -- ctx_wait in
            // pt_context.hip -- and the word must never go backwards under a later launch's wait)
            if (t <= drain_mark && drain_mark < t + c && S.L.drain_flag)
                (void)__hip_atomic_fetch_max(S.L.drain_flag, S.L.drain_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        have_at = __builtin_amdgcn_readfirstlane(t);
        have = c;
    };
    int cur_tile = 0, cur_used = 64, cur_fslot = 0, cur_ftile = 0;
    if constexpr (TUNED) draw(load_const_block(L.service)); else draw(S0);
    bool feed_empty = false;

    for (;;) {
        // walk phase: an inner loop of its own, so that the path state the walk does not
        // touch stays put in its registers
        bool serviceable;
        int nwalk = 0;
        for (;;) {
        const unsigned long long walking = __ballot(mode == M_TRAV);
        // a lane that is not walking waits for shading, for its next camera path or (while jobs are left) for a new job
        const unsigned long long waiting = __ballot(mode == M_SHADE || mode == M_PATH);
        serviceable = feed_empty ? (waiting != 0ull) : (walking != ~0ull);
        nwalk = (int)__popcll(walking);
        if (feed_empty && (k_tail_policy & 2)) {
            // Drain (no jobs left, the wave only finishes the paths it holds): the walk_min rule would
            // run a service step -- 2.4 node steps long -- for every single lane that ends a segment.
            // Waiting lanes are served once they are at least half as many as the walking ones.
            if (!(nwalk > 0 && 2 * (int)__popcll(waiting) < nwalk)) break;
        } else
        if (!(nwalk > 0 && (nwalk >= k_walk_min || !serviceable))) break;
        if (DEFER) {
            // ---- deferred-leaf walk (scenes whose leaves may be tested in any order: SceneRefs::
            // leaf_cap > 0).  In the in-order walk below a step runs the box tests for ~39 lanes
            // AND the triangle test for ~11, one after the other.  Here a leaf whose box is hit is
            // parked in the lane's leaf list (LDS slots 31 downwards; the node stack grows from
            // slot 0) while the lane goes on descending; the triangle test runs as a step of its
            // own once enough lanes have a leaf parked (or nothing else is left to do).  The set
            // of boxes and triangles tested is the one of the reference walk (it has no culling),
            // so the counters are unchanged, and the closest hit is the same because equal-t hits
            // are resolved by the leaf's rank in the reference's visiting order (strict '<' there
            // keeps the first visited, raytrace.wgsl:180).
            const bool trav = mode == M_TRAV;
            // (culling walks: a lane that is not walking has sp == 0 and nl == 0 -- both are zero before the first segment, a
            // segment ends when both are zero, and the 64-entry abort, which would leave entries behind, cannot fire there --
            // so the votes below are single comparisons, which the compiler turns into the lane mask directly; a vote on a
            // conjunction costs two more vector instructions to re-materialise the mask)
            const bool has_node = CULL ? sp > 0 : (trav && sp > 0), has_leaf = CULL ? nl > 0 : (trav && nl > 0);
            const int n_node = (int)__popcll(__ballot(has_node)), n_leaf = (int)__popcll(__ballot(has_leaf));
            const bool full = __ballot(CULL ? nl > lcap - (W8 ? 1 : (WIDE ? 4 : 2)) : (trav && nl > lcap - (WIDE ? 4 : 2))) != 0ull;      // a node step may park two (WIDE: four; 8-wide: one leaf ENTRY) more
            // (drain: whichever kind of step serves more lanes -- waiting for n_node == 0 would leave the lanes
            // that only have leaves idle for as long as the slowest descent takes)
            if (full || n_node == 0 || n_leaf >= k_leaf_min || (feed_empty && (k_tail_policy & 1) && n_leaf >= n_node)) {
                // every lane with a parked leaf tests one -- or two, when it has two (L.tri_pair): the second triangle's
                // loads are in flight with the first's, and the lane needs one triangle step less
                const bool two = k_tri_pair && (CULL ? nl > 1 : (has_leaf && nl > 1));
                if (!CW) u_tri += (uint32_t)n_leaf + (k_tri_pair ? (uint32_t)__popcll(__ballot(two)) : 0u);      // (wave-uniform count: scalar; CW counts per lane: only the leaves whose own box passes)
                if (wave_times) { st_switch(1); st_parked += wave_sum((uint32_t)(has_leaf ? nl : 0)); if (feed_empty) { st_tail_tri++; st_tail_lanes += (uint32_t)n_leaf; } }
                if (count_on) { st_tri_steps++; st_leaf_lanes += (uint32_t)n_leaf; }
                if constexpr (CW) {
                  if (has_leaf) {
                    // compressed-wide walk: a parked leaf is a CANDIDATE (its packet's box was rounded outward); the reference tests the
                    // triangle iff the leaf's own box passes its exact test -- made here, from the 64-byte record that carries that box
                    uint32_t ti, tj;
                    bool two_;
                    if constexpr (W8) {
                        // 8-wide walk: the top leaf entry {record base, hit slots in visiting order}: the nearest one or two of them
                        const int li = DEPTH - nl;
                        const uint32_t le = stack[li * 64];
                        uint32_t lh = le & 0xffu;
                        const uint32_t rb = le >> 8;
                        const uint32_t oct = (pre.flags >> 4) & 7u;
                        uint32_t pb = 31u - (uint32_t)__clz((int)lh);          // (hits in visiting order: the highest bit is the nearest slot)
                        ti = rb + (pb ^ oct);
                        lh &= ~(1u << pb);
                        two_ = k_tri_pair && lh != 0u;
                        tj = ti;
                        if (two_) { pb = 31u - (uint32_t)__clz((int)lh); tj = rb + (pb ^ oct); lh &= ~(1u << pb); }
                        stack[li * 64] = (le & 0xffffff00u) | lh;
                        nl -= lh == 0u ? 1 : 0;
                    } else {
                        two_ = two;
                        nl--;
                        ti = stack[(DEPTH - 1 - nl) * 64] & 0x7fffffffu;
                        tj = ti;
                        if (two_) { nl--; tj = stack[(DEPTH - 1 - nl) * 64] & 0x7fffffffu; }
                    }
                    const float4 *const TR = W8 ? sc.tripk8 : sc.tripk64;
                    float4 pa = TR[(size_t)ti * 4 + 0], pb = TR[(size_t)ti * 4 + 1], pc = TR[(size_t)ti * 4 + 2], pd = TR[(size_t)ti * 4 + 3];
                    // (the second triangle's registers are only read under `two`.  Left unset otherwise, and with whole-tuple barriers below,
                    // the step loses ~40 register moves -- 15 to zero-fill them, 15 to copy the loaded values into the filled registers, a dozen
                    // out of the first triangle's load tuples -- where the instantiation has registers for that: the one-axis-culling builds
                    // (YMAX).  The others, which carry the per-axis entry distances of four children through the node step, spill 2 .. 6
                    // registers with either change and keep the zero fill and the per-dword barriers; forming their culling value inside the
                    // box test to free those twelve registers was tried and costs the forest 11 %: profiles/r04_w_ab_tuple_barriers.log)
                    float4 qa, qb, qc, qd;
                    if constexpr (!YMAX || W8) { qa = make_float4(0.0f, 0.0f, 0.0f, 0.0f); qb = qa; qc = qa; qd = qa; }
                    if (two_) { qa = TR[(size_t)tj * 4 + 0]; qb = TR[(size_t)tj * 4 + 1]; qc = TR[(size_t)tj * 4 + 2]; qd = TR[(size_t)tj * 4 + 3]; }
                    keep16<YMAX && !W8>(pa);
                    keep16<YMAX && !W8>(pb);
                    keep16<YMAX && !W8>(pc);
                    keep16<YMAX && !W8>(pd);
                    {
                        float t, u, v;
                        // (8-wide walk: the record's last word = guard flag << 31 | the triangle's index; its own index is a record slot)
                        const uint32_t oi = W8 ? (__float_as_uint(pd.w) & 0x7fffffffu) : ti;
                        const bool inbox = leaf_box_hit(o, d, pre, W8 ? (__float_as_uint(pd.w) >> 31) != 0u : __float_as_uint(pd.w) != 0u, F3(pc.y, pc.z, pc.w), F3(pd.x, pd.y, pd.z));
                        const bool hit = ray_triangle_flat_e(o, d, F3(pa.x, pa.y, pa.z), F3(pa.w, pb.x, pb.y), F3(pb.z, pb.w, pc.x), t, u, v) && inbox;
                        cnt.tri += inbox ? 1u : 0u;
                        bool take = hit && t < best.t;
                        if (hit && t == best.t && best.tri >= 0) take = sc.leaf_rank[oi] < sc.leaf_rank[best.tri];
                        best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                        best.tri = take ? (int32_t)oi : best.tri;
                    }
                    if (two_) {
                        keep16<YMAX && !W8>(qa);
                        keep16<YMAX && !W8>(qb);
                        keep16<YMAX && !W8>(qc);
                        keep16<YMAX && !W8>(qd);
                        float t, u, v;
                        const uint32_t oj = W8 ? (__float_as_uint(qd.w) & 0x7fffffffu) : tj;
                        const bool inbox = leaf_box_hit(o, d, pre, W8 ? (__float_as_uint(qd.w) >> 31) != 0u : __float_as_uint(qd.w) != 0u, F3(qc.y, qc.z, qc.w), F3(qd.x, qd.y, qd.z));
                        const bool hit = ray_triangle_flat_e(o, d, F3(qa.x, qa.y, qa.z), F3(qa.w, qb.x, qb.y), F3(qb.z, qb.w, qc.x), t, u, v) && inbox;
                        cnt.tri += inbox ? 1u : 0u;
                        bool take = hit && t < best.t;
                        if (hit && t == best.t && best.tri >= 0) take = sc.leaf_rank[oj] < sc.leaf_rank[best.tri];
                        best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                        best.tri = take ? (int32_t)oj : best.tri;
                    }
                    if (sp == 0 && nl == 0) mode = M_SHADE;
                  }
                } else
                if (has_leaf) {
                    if (DIAG) lane_cost += two ? 6u : 3u;
                    nl--;
                    const uint32_t ti = stack[(DEPTH - 1 - nl) * 64] & 0x7fffffffu;
                    uint32_t tj = ti;
                    if (two) { nl--; tj = stack[(DEPTH - 1 - nl) * 64] & 0x7fffffffu; }
                    // (the opaque statements keep each of these ONE 16-byte load: left alone, the compiler fetches the nine
                    // coordinates as five overlapping 8- and 12-byte pieces to feed packed multiplies -- five divergent
                    // requests per lane instead of three)
                    float4 pa = sc.tripk[(size_t)ti * 3 + 0];
                    float4 pb = sc.tripk[(size_t)ti * 3 + 1];
                    float4 pc = sc.tripk[(size_t)ti * 3 + 2];
                    float4 qa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), qb = qa, qc = qa;
                    if (two) {
                        qa = sc.tripk[(size_t)tj * 3 + 0];
                        qb = sc.tripk[(size_t)tj * 3 + 1];
                        qc = sc.tripk[(size_t)tj * 3 + 2];
                    }
                    asm volatile("" : "+v"(pa.x), "+v"(pa.y), "+v"(pa.z), "+v"(pa.w));
                    asm volatile("" : "+v"(pb.x), "+v"(pb.y), "+v"(pb.z), "+v"(pb.w));
                    asm volatile("" : "+v"(pc.x), "+v"(pc.y), "+v"(pc.z), "+v"(pc.w));
                    {
                        float t, u, v;
                        const bool hit = ray_triangle_flat_e(o, d, xyz(pa), xyz(pb), xyz(pc), t, u, v);
                        bool take = hit && t < best.t;
                        if (hit && t == best.t && best.tri >= 0)      // rare: the earlier leaf of the reference order wins
                            take = sc.leaf_rank[ti] < sc.leaf_rank[best.tri];
                        best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                        best.tri = take ? (int32_t)ti : best.tri;
                    }
                    if (two) {
                        asm volatile("" : "+v"(qa.x), "+v"(qa.y), "+v"(qa.z), "+v"(qa.w));
                        asm volatile("" : "+v"(qb.x), "+v"(qb.y), "+v"(qb.z), "+v"(qb.w));
                        asm volatile("" : "+v"(qc.x), "+v"(qc.y), "+v"(qc.z), "+v"(qc.w));
                        float t, u, v;
                        const bool hit = ray_triangle_flat_e(o, d, xyz(qa), xyz(qb), xyz(qc), t, u, v);
                        bool take = hit && t < best.t;
                        if (hit && t == best.t && best.tri >= 0)
                            take = sc.leaf_rank[tj] < sc.leaf_rank[best.tri];
                        best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
                        best.tri = take ? (int32_t)tj : best.tri;
                    }
                    if (sp == 0 && nl == 0) mode = M_SHADE;
                }
            } else {
                if (!WIDE) u_box += 2u * (uint32_t)n_node;  // proper tree: both children of every popped node are tested (WIDE: per lane)
                if (wave_times) { st_switch(0); if (feed_empty) { st_tail_node++; st_tail_lanes += (uint32_t)n_node; } }
                if (count_on) { st_walk_steps++; st_walk_lanes += (uint32_t)n_node; }
                // (CULL) can any lane's node entries leave the LDS part of its stack in this step?  One pop, then up to two / four pushes.
                const bool shallow = CULL && __ballot(W8 ? sp >= NCAP8 : sp > NCAP - (WIDE ? 3 : 1)) == 0ull;
                if constexpr (W8) {
                  if (has_node) {
                    // ---- pop: the top entry names the internal children of one packet still to visit {first child packet; hits in visiting
                    // order (bits 0-7) | internal-slot mask (bits 8-15)}: the nearest is the highest bit, its slot that bit ^ octant, its packet
                    // the first child + the number of internal slots below it
                    const int top = sp - 1;
                    uint32_t e0, e1;
                    if (shallow) { e0 = stack[(2 * top) * 64]; e1 = stack[(2 * top + 1) * 64]; }
                    else {
                        const int tl = top < NCAP8 ? top : 0;
                        e0 = stack[(2 * tl) * 64]; e1 = stack[(2 * tl + 1) * 64];
                        asm volatile("" : "+v"(e0), "+v"(e1));          // keep these ds_reads of their own
                        if (top >= NCAP8) { e0 = ovf[(2 * (top - NCAP8)) * 64]; e1 = ovf[(2 * (top - NCAP8) + 1) * 64]; }
                    }
                    const uint32_t oct = (pre.flags >> 4) & 7u;
                    const uint32_t pbit = 31u - (uint32_t)__clz((int)(e1 & 0xffu));
                    const uint32_t slot8 = pbit ^ oct;
                    const uint32_t e1n = e1 & ~(1u << pbit);
                    const bool more = (e1n & 0xffu) != 0u;
                    const uint32_t node = e0 + (uint32_t)__popc((e1 >> 8) & ((1u << slot8) - 1u));
                    if (shallow) stack[(2 * top + 1) * 64] = e1n;            // (when nothing is left the slot is free: written all the same)
                    else if (more) { if (top < NCAP8) stack[(2 * top + 1) * 64] = e1n; else ovf[(2 * (top - NCAP8) + 1) * 64] = e1n; }
                    sp = more ? sp : top;
                    const float4 *P = sc.cw8 + (size_t)node * 5;
                    float4 c0 = P[0], c1 = P[1], c2 = P[2], c3 = P[3], c4 = P[4];
                    const uint32_t meta = __float_as_uint(c0.w), cb = __float_as_uint(c4.z), tb = __float_as_uint(c4.w);
                    const uint32_t imn = meta >> 24;
                    uint32_t mask = 0u;
                    const float cx = __uint_as_float((meta & 0xffu) << 23), cy = __uint_as_float(((meta >> 8) & 0xffu) << 23), cz = __uint_as_float(((meta >> 16) & 0xffu) << 23);
                    if ((pre.flags & 8u) == 0u) {
                        const float Ax = (c0.x - o.x) * pre.ix, Ay = (c0.y - o.y) * pre.iy, Az = (c0.z - o.z) * pre.iz;
                        const float Bx = cx * pre.ix, By = cy * pre.iy, Bz = cz * pre.iz;
                        // near and far plane per axis by the sign of the ray's reciprocal (see the 4-ary compressed walk below): twelve selects
                        // on the packed index words for the eight children
                        const bool nx_ = pre.ix < 0.0f, ny_ = pre.iy < 0.0f, nz_ = pre.iz < 0.0f;
                        const uint32_t lx0 = __float_as_uint(c1.x), lx1 = __float_as_uint(c1.y), ly0 = __float_as_uint(c1.z), ly1 = __float_as_uint(c1.w);
                        const uint32_t lz0 = __float_as_uint(c2.x), lz1 = __float_as_uint(c2.y), hx0 = __float_as_uint(c2.z), hx1 = __float_as_uint(c2.w);
                        const uint32_t hy0 = __float_as_uint(c3.x), hy1 = __float_as_uint(c3.y), hz0 = __float_as_uint(c3.z), hz1 = __float_as_uint(c3.w);
                        const uint32_t ex0 = nx_ ? hx0 : lx0, ex1 = nx_ ? hx1 : lx1, fx0 = nx_ ? lx0 : hx0, fx1 = nx_ ? lx1 : hx1;
                        const uint32_t ey0 = ny_ ? hy0 : ly0, ey1 = ny_ ? hy1 : ly1, fy0 = ny_ ? ly0 : hy0, fy1 = ny_ ? ly1 : hy1;
                        const uint32_t ez0 = nz_ ? hz0 : lz0, ez1 = nz_ ? hz1 : lz1, fz0 = nz_ ? lz0 : hz0, fz1 = nz_ ? lz1 : hz1;
                        // distance bound per child: W_k = wq_k * 2^(wexp - 127), one conversion and one product per child
                        const uint32_t wq0 = __float_as_uint(c4.x), wq1 = __float_as_uint(c4.y);
                        const float rcw = fmaf(cull_ka, best.t, cull_kb) * __uint_as_float((cb & 0xff000000u) >> 1);
                        const float bt = best.t * 1.00000095367431640625f;
                        const float aix = fabsf(pre.ix), aiy = fabsf(pre.iy), aiz = fabsf(pre.iz);
                        const float ymax = fmaxf(fmaxf(aix, aiy), aiz);
                        const float rcwy = rcw * ymax;
#define PT_C8(K)                                                                                                              \
                        {                                                                                                      \
                            const uint32_t xe_ = (K) < 4 ? ex0 : ex1, xf_ = (K) < 4 ? fx0 : fx1, ye_ = (K) < 4 ? ey0 : ey1, yf_ = (K) < 4 ? fy0 : fy1; \
                            const uint32_t ze_ = (K) < 4 ? ez0 : ez1, zf_ = (K) < 4 ? fz0 : fz1, ww_ = (K) < 4 ? wq0 : wq1;   \
                            const float ax_ = fmaf((float)((xe_ >> (8 * ((K) & 3))) & 0xffu), Bx, Ax), bx_ = fmaf((float)((xf_ >> (8 * ((K) & 3))) & 0xffu), Bx, Ax); \
                            const float ay_ = fmaf((float)((ye_ >> (8 * ((K) & 3))) & 0xffu), By, Ay), by_ = fmaf((float)((yf_ >> (8 * ((K) & 3))) & 0xffu), By, Ay); \
                            const float az_ = fmaf((float)((ze_ >> (8 * ((K) & 3))) & 0xffu), Bz, Az), bz_ = fmaf((float)((zf_ >> (8 * ((K) & 3))) & 0xffu), Bz, Az); \
                            const float key_ = fmaxf(fmaxf(ax_, ay_), az_), f_ = fminf(fminf(bx_, by_), bz_);                 \
                            const float wk_ = (float)((ww_ >> (8 * ((K) & 3))) & 0xffu);                                       \
                            float tc_;                                                                                        \
                            if constexpr (YMAX) tc_ = fmaf(-wk_, rcwy, key_);                                                 \
                            else { const float dk_ = wk_ * rcw; tc_ = fmaxf(fmaxf(fmaf(-dk_, aix, ax_), fmaf(-dk_, aiy, ay_)), fmaf(-dk_, aiz, az_)); } \
                            mask |= (cwide_hit(key_, f_) && !(tc_ > bt)) ? (1u << (K)) : 0u;                                   \
                        }
                        PT_C8(0) PT_C8(1) PT_C8(2) PT_C8(3) PT_C8(4) PT_C8(5) PT_C8(6) PT_C8(7)
#undef PT_C8
                    } else {
                        // a ray on the plain-division path: the reference's test on the decoded boxes (conservative: see the 4-ary walk below);
                        // its culling constants are +inf, nothing is skipped.  Rare: a loop, not eight copies of the divisions
#pragma unroll 1
                        for (int k = 0; k < 8; k++) {
                            const uint32_t sh = 8u * (uint32_t)(k & 3);
                            const bool hi_ = k >= 4;
                            const uint32_t lxw = __float_as_uint(hi_ ? c1.y : c1.x), lyw = __float_as_uint(hi_ ? c1.w : c1.z), lzw = __float_as_uint(hi_ ? c2.y : c2.x);
                            const uint32_t hxw = __float_as_uint(hi_ ? c2.w : c2.z), hyw = __float_as_uint(hi_ ? c3.y : c3.x), hzw = __float_as_uint(hi_ ? c3.w : c3.z);
                            const bool h_ = ray_aabb(o, d, fmaf((float)((lxw >> sh) & 0xffu), cx, c0.x), fmaf((float)((lyw >> sh) & 0xffu), cy, c0.y), fmaf((float)((lzw >> sh) & 0xffu), cz, c0.z),
                                                     fmaf((float)((hxw >> sh) & 0xffu), cx, c0.x), fmaf((float)((hyw >> sh) & 0xffu), cy, c0.y), fmaf((float)((hzw >> sh) & 0xffu), cz, c0.z));
                            mask |= h_ ? (1u << k) : 0u;
                        }
                    }
                    cnt.box += (tb >> 24) & 15u;         // the packet's number of children
                    // the leaves that were hit: ONE entry {record base, hit slots} in the lane's leaf list (nl <= LCAP - 1 before the step: the slot is free)
                    // both hit masks -- internal children in bits 0-7, leaves in bits 8-15 -- moved from slot order into visiting order
                    // (bit s -> bit s ^ octant: three conditional swaps, on both bytes at once)
                    uint32_t hb = (mask & imn) | ((mask & ~imn & 0xffu) << 8);
                    hb = (oct & 1u) ? (((hb & 0x5555u) << 1) | ((hb >> 1) & 0x5555u)) : hb;
                    hb = (oct & 2u) ? (((hb & 0x3333u) << 2) | ((hb >> 2) & 0x3333u)) : hb;
                    hb = (oct & 4u) ? (((hb & 0x0f0fu) << 4) | ((hb >> 4) & 0x0f0fu)) : hb;
                    const uint32_t lh = hb >> 8, hp = hb & 0xffu;
                    stack[(DEPTH - 1 - nl) * 64] = (tb << 8) | lh;
                    nl += lh != 0u ? 1 : 0;
                    const uint32_t n0 = cb & 0xffffffu, n1 = hp | (imn << 8);
                    if (shallow) { stack[(2 * sp) * 64] = n0; stack[(2 * sp + 1) * 64] = n1; }
                    else if (hp != 0u) {
                        if (sp < NCAP8) { stack[(2 * sp) * 64] = n0; stack[(2 * sp + 1) * 64] = n1; }
                        else { ovf[(2 * (sp - NCAP8)) * 64] = n0; ovf[(2 * (sp - NCAP8) + 1) * 64] = n1; }
                    }
                    sp += hp != 0u ? 1 : 0;
                    if (sp == 0 && nl == 0) mode = M_SHADE;
                  }
                } else
                if (WIDE) {
                  if (has_node) {
                    if (DIAG) lane_cost += 4u;
                    const uint32_t ref = shallow ? flat_pop() : cull_pop();
                    uint32_t cr[4], w01, w23, nchild;
                    bool hit[4];
                    float key[4];                    // entry distance of each box (approximate with FILT): sort key and culling bound
                    f3 tn[4];                        // per-axis entry distances (!YMAX)
                    // (a ray on the plain-division path leaves them at -PT_INF: never skipped, order of no consequence; the compressed-wide
                    // walk: see that branch)
                    if constexpr (!(CW && !YMAX)) {
                        tn[0] = tn[1] = tn[2] = tn[3] = F3(-PT_INF, -PT_INF, -PT_INF);
                        key[0] = key[1] = key[2] = key[3] = -PT_INF;
                    }
                    if constexpr (CW) {
                        // compressed wide packet: 64 bytes, boxes on the node's 8-bit grid, rounded outward (CWidePacket; cwide_hit)
                        const float4 *P = sc.cwide + (size_t)ref * 4;
                        float4 c0, c1, c2, c3;
                        if (TOPLDS && ref < ntop) {       // top of the tree: this wave's LDS copy
                            c0 = top_lds[ref * 4 + 0]; c1 = top_lds[ref * 4 + 1]; c2 = top_lds[ref * 4 + 2]; c3 = top_lds[ref * 4 + 3];
                            asm volatile("" : "+v"(c0.x), "+v"(c1.x), "+v"(c2.x), "+v"(c3.x));   // keep these ds_read_b128
                        } else { c0 = P[0]; c1 = P[1]; c2 = P[2]; c3 = P[3]; }
                        cr[0] = __float_as_uint(c3.x); cr[1] = __float_as_uint(c3.y); cr[2] = __float_as_uint(c3.z); cr[3] = __float_as_uint(c3.w);
                        const uint32_t meta = __float_as_uint(c0.w);
                        w01 = __float_as_uint(c2.z); w23 = __float_as_uint(c2.w); nchild = (meta >> 24) & 7u;
                        const uint32_t lx = __float_as_uint(c1.x), ly = __float_as_uint(c1.y), lz = __float_as_uint(c1.z);
                        const uint32_t hx = __float_as_uint(c1.w), hy = __float_as_uint(c2.x), hz = __float_as_uint(c2.y);
                        const float cx = __uint_as_float((meta & 0xffu) << 23), cy = __uint_as_float(((meta >> 8) & 0xffu) << 23), cz = __uint_as_float(((meta >> 16) & 0xffu) << 23);
                        if ((pre.flags & 8u) == 0u) {
                            const float Ax = (c0.x - o.x) * pre.ix, Ay = (c0.y - o.y) * pre.iy, Az = (c0.z - o.z) * pre.iz;
                            const float Bx = cx * pre.ix, By = cy * pre.iy, Bz = cz * pre.iz;
                            // near and far plane per axis by the SIGN of the ray's reciprocal: with B >= 0 the quotient fma(q, B, A) rises with
                            // the index q, with B < 0 it falls, and neither quotient is a NaN here (A and q B are finite: coordinates below 1e30,
                            // |1/d| <= 1e6) -- so min(a, b) / max(a, b) of the two planes' quotients ARE the quotient of the lower / upper index
                            // (B >= 0) or the other way round, bit for bit; chosen once per axis for the four children (six selects on the packed
                            // index words) instead of a min and a max per axis and child (24)
                            const bool nx_ = pre.ix < 0.0f, ny_ = pre.iy < 0.0f, nz_ = pre.iz < 0.0f;
                            const uint32_t ex = nx_ ? hx : lx, fx = nx_ ? lx : hx, ey = ny_ ? hy : ly, fy = ny_ ? ly : hy, ez = nz_ ? hz : lz, fz = nz_ ? lz : hz;
#define PT_CBOX(K)                                                                                                          \
                            {                                                                                              \
                                const float ax_ = fmaf((float)((ex >> (8 * K)) & 0xffu), Bx, Ax), bx_ = fmaf((float)((fx >> (8 * K)) & 0xffu), Bx, Ax); \
                                const float ay_ = fmaf((float)((ey >> (8 * K)) & 0xffu), By, Ay), by_ = fmaf((float)((fy >> (8 * K)) & 0xffu), By, Ay); \
                                const float az_ = fmaf((float)((ez >> (8 * K)) & 0xffu), Bz, Az), bz_ = fmaf((float)((fz >> (8 * K)) & 0xffu), Bz, Az); \
                                const f3 a_ = F3(ax_, ay_, az_);                                                           \
                                const float f_ = fminf(fminf(bx_, by_), bz_);        /* (no NaN among them: min(INF, .) drops out) */ \
                                key[K] = fmaxf(fmaxf(a_.x, a_.y), a_.z);                                                   \
                                if constexpr (!YMAX) tn[K] = a_;                                                           \
                                hit[K] = cwide_hit(key[K], f_);                                                            \
                            }
                            PT_CBOX(0) PT_CBOX(1) PT_CBOX(2) PT_CBOX(3)
#undef PT_CBOX
                        } else {
                            // (builds without the one-axis culling condition: its sort keys and entry distances are left UNSET -- an empty statement
                            // that "defines" them, no instruction; with that condition the four keys keep their -PT_INF from above: there the
                            // unset form spills two registers.  Such
                            // a ray's culling constants are +inf (cull_setup), so that its culling value is -inf or a NaN whatever these hold
                            // and no child is skipped; the order of its children is of no consequence.  Set to -PT_INF here or up front they
                            // are 4 (16 without the one-axis culling condition) moves in EVERY node step: the compiler puts a branch's constant
                            // assignments where both branches meet)
                            if constexpr (!YMAX) {
                                asm("" : "=v"(key[0]), "=v"(key[1]), "=v"(key[2]), "=v"(key[3]));
                                asm("" : "=v"(tn[0].x), "=v"(tn[0].y), "=v"(tn[0].z), "=v"(tn[1].x), "=v"(tn[1].y), "=v"(tn[1].z),
                                         "=v"(tn[2].x), "=v"(tn[2].y), "=v"(tn[2].z), "=v"(tn[3].x), "=v"(tn[3].y), "=v"(tn[3].z));
                            }
                            // a ray on the plain-division path (a parallel axis, an out-of-range component): the reference's test on the
                            // decoded box.  The decoded planes RN(o + cell q) are NOT exact in general (o is not a multiple of the cell), but
                            // round-to-nearest is monotone and the builder moved every plane outward by at least a whole cell, so the decoded
                            // box still contains the child's (prepare_cull verifies exactly this fma for every plane of every packet) -- and
                            // the reference's test is monotone under containment, hence conservative too
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                hit[k] = ray_aabb(o, d, fmaf((float)((lx >> (8 * k)) & 0xffu), cx, c0.x), fmaf((float)((ly >> (8 * k)) & 0xffu), cy, c0.y), fmaf((float)((lz >> (8 * k)) & 0xffu), cz, c0.z),
                                                  fmaf((float)((hx >> (8 * k)) & 0xffu), cx, c0.x), fmaf((float)((hy >> (8 * k)) & 0xffu), cy, c0.y), fmaf((float)((hz >> (8 * k)) & 0xffu), cz, c0.z));
                        }
                    } else {
                    const float4 *P = sc.wide + (size_t)ref * 8;
                    const float4 q0 = P[0], q1 = P[1], q2 = P[2], q3 = P[3], q4 = P[4], q5 = P[5], q6 = P[6], q7 = P[7];
                    cr[0] = __float_as_uint(q6.x); cr[1] = __float_as_uint(q6.y); cr[2] = __float_as_uint(q6.z); cr[3] = __float_as_uint(q6.w);
                    const uint32_t wf = __float_as_uint(q7.z);
                    w01 = __float_as_uint(q7.x); w23 = __float_as_uint(q7.y); nchild = (wf >> 4) & 7u;
                    if (((pre.flags & 8u) | (wf & 15u)) == 0u) {
                        if constexpr (FILT) {
                            // filtered slab test (see slab_q0): one multiplication per quotient; box by box, the exact test of an
                            // undecided box right behind its approximate one, while the box's six coordinates are still in
                            // registers (fetching them again costs the step a second memory round trip: measured)
#define PT_BOX(K, MNX, MNY, MNZ, MXX, MXY, MXZ)                                                              \
                            {                                                                                \
                                f3 a_;                                                                       \
                                float f_;                                                                    \
                                slab_q0(o, pre, MNX, MNY, MNZ, MXX, MXY, MXZ, a_, f_);                       \
                                key[K] = fmaxf(fmaxf(a_.x, a_.y), a_.z);                                     \
                                if constexpr (!YMAX) tn[K] = a_;                                             \
                                hit[K] = slab_hit(key[K], f_);                                               \
                                if (!(slab_margin(key[K], f_) > 0.0f))                                       \
                                    hit[K] = ray_aabb_fast(o, d, pre, MNX, MNY, MNZ, MXX, MXY, MXZ);         \
                            }
                            PT_BOX(0, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y)
                            PT_BOX(1, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w)
                            PT_BOX(2, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y)
                            PT_BOX(3, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w)
#undef PT_BOX
                        } else {
                            hit[0] = ray_aabb_fast_t(o, d, pre, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn[0]);
                            hit[1] = ray_aabb_fast_t(o, d, pre, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tn[1]);
                            hit[2] = ray_aabb_fast_t(o, d, pre, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, tn[2]);
                            hit[3] = ray_aabb_fast_t(o, d, pre, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, tn[3]);
#pragma unroll
                            for (int k = 0; k < 4; k++) key[k] = fmaxf(fmaxf(tn[k].x, tn[k].y), tn[k].z);
                        }
                    } else {
                        hit[0] = ray_aabb_pre(o, d, pre, (wf & 1u) != 0u, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
                        hit[1] = ray_aabb_pre(o, d, pre, (wf & 2u) != 0u, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
                        hit[2] = ray_aabb_pre(o, d, pre, (wf & 4u) != 0u, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y);
                        hit[3] = ray_aabb_pre(o, d, pre, (wf & 8u) != 0u, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w);
                    }
                    }
                    // distance bound per child, entry distances as sort keys
                    const float rc = fmaf(cull_ka, best.t, cull_kb);
                    const float bt = best.t * 1.00000095367431640625f;
                    const float wgt[4] = { __uint_as_float(w01 & 0xffff0000u), __uint_as_float(w01 << 16),
                                           __uint_as_float(w23 & 0xffff0000u), __uint_as_float(w23 << 16) };
                    // YMAX: (S) on the axis that sets the entry distance, with |RN(1/d_i)| replaced by the largest of the
                    // three -- a weaker condition than (S) on that axis, hence still sufficient
                    const float ymax = fmaxf(fmaxf(fabsf(pre.ix), fabsf(pre.iy)), fabsf(pre.iz));
                    // (one axis: W_k (rc ymax) with the product of the two per-step factors formed ONCE -- one multiplication per child less than
                    // (W_k rc) ymax, the same two roundings: PROOFS.md 1 budgets six for the left side of (S))
                    const float rcy = rc * ymax;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float tc;
                        if constexpr (YMAX) tc = fmaf(-wgt[k], rcy, key[k]);
                        else {
                            const float dk = wgt[k] * rc;
                            tc = fmaxf(fmaxf(fmaf(-dk, fabsf(pre.ix), tn[k].x), fmaf(-dk, fabsf(pre.iy), tn[k].y)), fmaf(-dk, fabsf(pre.iz), tn[k].z));
                        }
                        // a child that is missed or skipped becomes an empty entry: the sort below then moves (key, reference)
                        // pairs only -- selects, no branches, no lane masks to swap
                        cr[k] = (hit[k] && !(tc > bt)) ? cr[k] : PT_REF_NONE;
                    }
                    cnt.box += nchild;           // the packet's number of children
                    // far first, near last (popped first): sort the four entries by entry distance, descending
#define PT_CSWAP(A, B)                                                                         \
                    {                                                                          \
                        const bool sw = key[A] < key[B];                                       \
                        const float ka_ = sw ? key[B] : key[A], kb_ = sw ? key[A] : key[B];    \
                        const uint32_t ra_ = sw ? cr[B] : cr[A], rb_ = sw ? cr[A] : cr[B];     \
                        key[A] = ka_; key[B] = kb_; cr[A] = ra_; cr[B] = rb_;                  \
                    }
                    PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#undef PT_CSWAP
                    // nl <= LCAP - 4 before the step (the `full` rule): the four leaf slots are free
                    if (shallow) {
#pragma unroll
                        for (int k = 0; k < 4; k++) flat_push(cr[k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) cull_push(cr[k], cr[k] != PT_REF_NONE);
                    }
                    if (sp == 0 && nl == 0) mode = M_SHADE;
                  }
                } else
                if (has_node) {
                    if (DIAG) lane_cost += 3u;        // (a binary packet: two boxes)
                    uint32_t ref;
                    if (CULL) ref = shallow ? flat_pop() : cull_pop();
                    else { sp--; ref = stack[sp * 64]; }
                    float4 p0, p1, p2, p3;
                    if (TOPLDS && ref < ntop) {       // top of the tree: this wave's LDS copy
                        p0 = top_lds[ref * 4 + 0]; p1 = top_lds[ref * 4 + 1];
                        p2 = top_lds[ref * 4 + 2]; p3 = top_lds[ref * 4 + 3];
                        asm volatile("" : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x));   // keep these ds_read_b128
                    } else {
                        p0 = sc.packets[(size_t)ref * 4 + 0]; p1 = sc.packets[(size_t)ref * 4 + 1];
                        p2 = sc.packets[(size_t)ref * 4 + 2]; p3 = sc.packets[(size_t)ref * 4 + 3];
                    }
                    const uint32_t lref = __float_as_uint(p3.x), rref = __float_as_uint(p3.y);
                    const uint32_t pf = __float_as_uint(p3.z);
                    
                    bool hl, hr;
                    f3 nl3 = F3(-PT_INF, -PT_INF, -PT_INF), nr3 = nl3;       // CULL: per-axis entry distances (-INF: never skipped)
                    if (((pre.flags & 8u) | pf) == 0u) {
                        if (CULL) {
                            hl = ray_aabb_fast_t(o, d, pre, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, nl3);
                            hr = ray_aabb_fast_t(o, d, pre, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w, nr3);
                        } else {
                            hl = ray_aabb_fast(o, d, pre, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y);
                            hr = ray_aabb_fast(o, d, pre, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w);
                        }
                    } else {
                        hl = ray_aabb_pre(o, d, pre, (pf & 1u) != 0u, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y);
                        hr = ray_aabb_pre(o, d, pre, (pf & 2u) != 0u, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w);
                    }
                    uint32_t r1 = lref, r2 = rref;
                    if (CULL) {
                        // distance bound (see the kernel's header comment and DESIGN.md 3a): a hit below this
                        // child lies within delta of its box, so on EVERY axis it is at least
                        // tnear_i - delta / |d_i| away; skip when that exceeds the closest hit so far
                        const uint32_t pe = __float_as_uint(p3.w);
                        const float rc = fmaf(cull_ka, best.t, cull_kb);
                        const float dl = __uint_as_float(pe & 0xffff0000u) * rc, dr = __uint_as_float(pe << 16) * rc;
                        const float bt = best.t * 1.00000095367431640625f;       // 1 + 2^-20: the roundings of tnear and of the fma
                        const float tl = fmaxf(fmaxf(fmaf(-dl, fabsf(pre.ix), nl3.x), fmaf(-dl, fabsf(pre.iy), nl3.y)), fmaf(-dl, fabsf(pre.iz), nl3.z));
                        const float tr = fmaxf(fmaxf(fmaf(-dr, fabsf(pre.ix), nr3.x), fmaf(-dr, fabsf(pre.iy), nr3.y)), fmaf(-dr, fabsf(pre.iz), nr3.z));
                        // a child that is missed or skipped becomes an empty entry (selects: no lane masks to swap);
                        // far child first, near child last (popped first); leaves go to the leaf list anyway
                        const uint32_t el = (hl && !(tl > bt)) ? lref : PT_REF_NONE, er = (hr && !(tr > bt)) ? rref : PT_REF_NONE;
                        const bool sw = fmaxf(fmaxf(nl3.x, nl3.y), nl3.z) < fmaxf(fmaxf(nr3.x, nr3.y), nr3.z);
                        r1 = sw ? er : el; r2 = sw ? el : er;
                        hl = r1 != PT_REF_NONE; hr = r2 != PT_REF_NONE;
                    }
                    // sp + nl <= 30 here (leaf_cap = 32 - worst-case stack, nl <= leaf_cap - 2), so slot sp
                    // and slot 31 - nl are both free: the stores are unconditional, the counts select
                    if (CULL && shallow) {
                        flat_push(r1);
                        flat_push(r2);
                    } else if (CULL) {
                        cull_push(r1, hl);
                        cull_push(r2, hr);
                    } else {
                        const bool ll = (r1 & PT_REF_LEAF) != 0u, rl = (r2 & PT_REF_LEAF) != 0u;
                        stack[(ll ? DEPTH - 1 - nl : sp) * 64] = ll ? (r1 & 0x7fffffffu) : r1;
                        nl += (hl && ll) ? 1 : 0;
                        sp += (hl && !ll) ? 1 : 0;
                        stack[(rl ? DEPTH - 1 - nl : sp) * 64] = rl ? (r2 & 0x7fffffffu) : r2;
                        nl += (hr && rl) ? 1 : 0;
                        sp += (hr && !rl) ? 1 : 0;
                    }
                    if (sp == 0 && nl == 0) mode = M_SHADE;
                }
            }
            continue;
        }
        {
            // ---- walk step: one pop per walking lane (raytrace.wgsl:166-200)
            if (count_on) { st_walk_steps++; st_walk_lanes += (uint32_t)nwalk; }
            // Wave-uniform choice: when no walking lane needs the plain-division test or the
            // overflow part of the stack, the step runs a version without those branches
            // (stack accesses are plain LDS, both child boxes are tested and pushed without
            // divergence); packets with a guard bit take the generic child handling.
            const bool trav = mode == M_TRAV;
            if (__ballot(trav && ((pre.flags & 8u) != 0u || sp >= DEPTH)) == 0ull) {
                if (trav) {
                    sp--;
                    const uint32_t ref = stack[sp * 64];
                    if (wave_times) st_leaf_lanes += (uint32_t)__popcll(__ballot((ref & PT_REF_LEAF) != 0u));
                    if (ref & PT_REF_LEAF) {
                        const uint32_t ti = ref & 0x7fffffffu;
                        const float4 pa = sc.tripk[(size_t)ti * 3 + 0];
                        const float4 pb = sc.tripk[(size_t)ti * 3 + 1];
                        const float4 pc = sc.tripk[(size_t)ti * 3 + 2];
                        cnt.tri++;
                        float t, u, v;
                        if (ray_triangle_e(o, d, xyz(pa), xyz(pb), xyz(pc), t, u, v) && t < best.t) {
                            best.t = t; best.u = u; best.v = v; best.tri = (int32_t)ti;
                        }
                    } else {
                        const float4 p0 = sc.packets[(size_t)ref * 4 + 0], p1 = sc.packets[(size_t)ref * 4 + 1];
                        const float4 p2 = sc.packets[(size_t)ref * 4 + 2], p3 = sc.packets[(size_t)ref * 4 + 3];
                        const uint32_t lref = __float_as_uint(p3.x), rref = __float_as_uint(p3.y);
                        const uint32_t pf = __float_as_uint(p3.z);
                        if (pf == 0u) {
                            cnt.box += 2;
                            const bool hl = ray_aabb_fast(o, d, pre, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y);
                            const bool hr = ray_aabb_fast(o, d, pre, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w);
                            // sp <= PT_SM_LDS_DEPTH - 2 here: both slots exist; a slot above sp is dead
                            stack[sp * 64] = lref;
                            sp += hl ? 1 : 0;
                            stack[sp * 64] = rref;
                            sp += hr ? 1 : 0;
                        } else {
                            if (lref != PT_REF_NONE) {
                                cnt.box++;
                                if (ray_aabb_pre(o, d, pre, (pf & 1u) != 0u, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y)) { stack[sp * 64] = lref; sp++; }
                            }
                            if (rref != PT_REF_NONE) {
                                cnt.box++;
                                if (ray_aabb_pre(o, d, pre, (pf & 2u) != 0u, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w)) { stack[sp * 64] = rref; sp++; }
                            }
                        }
                    }
                    if (sp == 0) mode = M_SHADE;
                }
                continue;
            }
            if (mode == M_TRAV) {
                sp--;
                const uint32_t ref = st_load(sp);
                if (wave_times) st_leaf_lanes += (uint32_t)__popcll(__ballot((ref & PT_REF_LEAF) != 0u));
                if (ref & PT_REF_LEAF) {
                    const uint32_t ti = ref & 0x7fffffffu;
                    const float4 pa = sc.tripk[(size_t)ti * 3 + 0];
                    const float4 pb = sc.tripk[(size_t)ti * 3 + 1];
                    const float4 pc = sc.tripk[(size_t)ti * 3 + 2];
                    cnt.tri++;
                    float t, u, v;
                    if (ray_triangle_e(o, d, xyz(pa), xyz(pb), xyz(pc), t, u, v) && t < best.t) {
                        best.t = t; best.u = u; best.v = v; best.tri = (int32_t)ti;
                    }
                } else {
                    float4 p0, p1, p2, p3;
                    if (TOPLDS && ref < ntop) {       // top of the tree: LDS
                        p0 = top_lds[ref * 4 + 0]; p1 = top_lds[ref * 4 + 1];
                        p2 = top_lds[ref * 4 + 2]; p3 = top_lds[ref * 4 + 3];
                        asm volatile("" : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x));   // keep these ds_read_b128
                    } else {
                        p0 = sc.packets[(size_t)ref * 4 + 0]; p1 = sc.packets[(size_t)ref * 4 + 1];
                        p2 = sc.packets[(size_t)ref * 4 + 2]; p3 = sc.packets[(size_t)ref * 4 + 3];
                    }
                    const uint32_t lref = __float_as_uint(p3.x), rref = __float_as_uint(p3.y);
                    if (lref != PT_REF_NONE) {
                        cnt.box++;
                        if (ray_aabb_pre(o, d, pre, (__float_as_uint(p3.z) & 1u) != 0u, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y)) { st_store(sp, lref); sp++; }
                    }
                    if (rref != PT_REF_NONE) {
                        cnt.box++;
                        if (ray_aabb_pre(o, d, pre, (__float_as_uint(p3.z) & 2u) != 0u, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w)) { st_store(sp, rref); sp++; }
                    }
                }
                if (sp >= PT_MAX_STACK) { cnt.overflow++; mode = M_SHADE; }     // :167-171
                else if (sp == 0) mode = M_SHADE;
            }
        }
        }
        if (!serviceable) break;      // nobody walking, nothing waiting, no jobs left

        // ---- service step
        // Two groups wait here.  H: lanes whose walk ended in a hit (material fetch, 7 rand, cosine
        // direction: raytrace.wgsl:380-395).  B: lanes whose walk ended in a miss (equirect lookup,
        // :396-407), lanes between two camera paths, and free lanes while jobs are left (refill + the
        // camera ray, :441-455).  Served together, each half runs with the other half's lanes masked
        // off.  With L.shade_split > 0, and while other lanes are still walking, a step serves the
        // larger group and leaves the smaller one waiting (unless it has shade_split lanes by itself):
        // the groups alternate, each at a fuller mask.  Per-lane arithmetic is untouched: same bits.
        // (from here on `L`, `sc`, `un` and the derived scalars are this step's view of the RtService block)
        // (tuned: the block is loaded again at the start of each part of the step -- PT_SERVICE_PART -- so that only what
        // that part reads is in scalar registers at a time; loaded once for the whole step, ~70 values at once, the compiler
        // spilled half of them straight back to vector-register lanes)
        RtService *const service_block = L.service;
        RtService S;
        if constexpr (TUNED) S = load_const_block(service_block); else S = S0;
#define PT_SERVICE_PART() do { if constexpr (TUNED) S = load_const_block(service_block); } while (0)
        const RtLaunch &L = S.L;
        const SceneRefs &sc = L.scene;
        const RtUniforms &un = L.un;
        const CameraFrame &cf = S.cf;
        const f3 &cam_pos = S.cam_pos;
        const float &sinr = S.sinr, &cosr = S.cosr, &inv_res_x = S.inv_res_x, &inv_res_y = S.inv_res_y, &spf_f = S.spf_f;
        const uint32_t &res_w = S.res_w, &res_h = S.res_h;
        const int32_t &pinhole = S.pinhole, &res_ordinary_ = S.res_ordinary;
        const int &tiles_x = S.tiles_x, &ntiles_frame = S.ntiles_frame, &ntiles = S.ntiles;
        const int &grp_jobs = S.grp_jobs, &grp_full = S.grp_full, &grp_last = S.grp_last;
        const FastDiv &dv_frame = S.dv_frame, &dv_grp = S.dv_grp, &dv_gs = S.dv_gs, &dv_last = S.dv_last, &dv_tx = S.dv_tx;
        const unsigned long long m_hit = __ballot(mode == M_SHADE && best.tri >= 0);
        const int n_hit = (int)__popcll(m_hit);
        const int n_b = (int)__popcll(__ballot((mode == M_SHADE && best.tri < 0) || mode == M_PATH || (mode == M_DEAD && !feed_empty)));
        bool do_hit = n_hit > 0, do_b = true;
        if (k_shade_split > 0 && nwalk > 0 && n_hit > 0 && n_b > 0 && !(feed_empty && (k_tail_policy & 4))) {
            if (n_hit >= n_b) do_b = n_b >= k_shade_split;
            else do_hit = n_hit >= k_shade_split;
        }
        if (wave_times) {
            st_switch(2);
            if (feed_empty) st_tail_service++;
        }
        if (count_on) {
            st_service_steps++;
            if (do_hit) { st_hit_steps++; st_hit_lanes += (uint32_t)n_hit; }
            if (do_b) { st_b_steps++; st_shade_lanes += (uint32_t)__popcll(__ballot(mode == M_SHADE && best.tri < 0)); }
        }
        bool need_segment = false;    // start rayBVHIntersect for (o, d)
        {
            const bool shade_hit = do_hit && mode == M_SHADE && best.tri >= 0;
            const bool shade_miss = do_b && mode == M_SHADE && best.tri < 0;
            u_hit += (uint32_t)__popcll(__ballot(shade_hit));
            u_miss += (uint32_t)__popcll(__ballot(shade_miss));
            bool ended = true;
            if (shade_hit || shade_miss) {
                if constexpr (PARKG) {
                    ray_color = F3(parkg[0], parkg[64], parkg[128]);
                    light = F3(parkg[192], parkg[256], parkg[320]);
                } else {
                    ray_color = F3(park[0], park[64], park[128]);
                    light = F3(park[192], park[256], park[320]);
                }
            }
#ifdef PT_DIAG_SERVICE
            if (wave_times) st_switch(3);
#endif
            if (shade_hit) {          // trace(), raytrace.wgsl:380-395
                
                // finish_hit() (raytrace.wgsl:105-112) taken apart so that the two dependent round trips -- the triangle's normals and
                // material index, then the material -- are in flight while work that needs neither runs: the normals are requested
                // first, randDirection's ~300 instructions run, the material is requested as soon as its index is back, and the
                // normal's interpolation runs while that is in flight.  Same operations on the same values in the seed's order.
                float4 q3 = sc.tris[(size_t)best.tri * 7 + 3];
                float4 q4 = sc.tris[(size_t)best.tri * 7 + 4];
                float4 q5 = sc.tris[(size_t)best.tri * 7 + 5];
                const f3 rdir = rand_direction(seed);
                asm volatile("" : "+v"(q5.w));         // (the index is waited for HERE, not where the loads were issued, nor later)
                const int32_t mi = __float_as_int(q5.w);
                const float4 m0 = sc.mats[(size_t)mi * 4 + 0];
                const float4 m1 = sc.mats[(size_t)mi * 4 + 1];
                const float4 m2 = sc.mats[(size_t)mi * 4 + 2];
                const float4 m3 = sc.mats[(size_t)mi * 4 + 3];
                const float w = 1.0f - best.u - best.v;
                const f3 position = o + d * best.t;
                const f3 normal = normalize((xyz(q3) * w + xyz(q4) * best.u) + xyz(q5) * best.v);
                const f3 diffuse_dir = normalize(normal + rdir);
                const f3 specular_dir = reflect(d, normal);
                float is_specular = 0.0f;
                if (m2.x >= rand1(seed)) is_specular = 1.0f;
                o = position;
                d = mix(diffuse_dir, specular_dir, is_specular * (1.0f - m1.w));
                const f3 emitted = xyz(m3) * m3.w;
                light = light + emitted * ray_color;
                ray_color = ray_color * mix(xyz(m0), xyz(m1), is_specular);
                slot++;
                ended = (int32_t)(slot & 0xffffu) >= un.max_bounces;
            }
#ifdef PT_DIAG_SERVICE
            if (wave_times) st_switch(4);
#endif
            PT_SERVICE_PART();
            if (shade_miss) {         // :396-407
                
                float u, v;
                env_uv_from_dir(d, sinr, cosr, u, v);
                const f3 env = sample_env(sc.env, ENV_W, ENV_H, u, v);      // (the environment texture is 1024 x 512 by the API: renderer.ts:76-85, mi3pt_upload_environment)
                light = light + (ray_color * env) * un.env_intensity;
            }
            PT_SERVICE_PART();
            if (shade_hit || shade_miss) {
                mode = M_DEAD;        // until a path / segment is started below
                if (ended) {
                    {
                        // the sample is finished: incomingLight += trace(...) (:450); after the last one the pixel is, too (:455, :477).
                        // One sample per frame (the reference's default): incomingLight = 0 + light, stored at once.  More: the
                        // running sum and the number of samples taken rest in the pixel's own texel of the frame's radiance slot
                        // (xyz, w; zeroed when the job was handed out) -- the lane is the texel's only writer until the pixel is
                        // finished, and nothing has to be carried through the walks.  Same additions in the same order.
                        f3 pix = F3(0.0f, 0.0f, 0.0f) + light;
                        bool finished = true;
                        if (un.samples_per_frame != 1) {
                            float4 *const px = L.radiance + ((size_t)(slot >> 16) * L.slot_pixels + gx);
                            const float4 sum = *px;
                            pix = xyz(sum) + light;
                            const float taken = sum.w + 1.0f;
                            finished = taken >= spf_f;
                            if (finished) pix = F3(pix.x / spf_f, pix.y / spf_f, pix.z / spf_f);
                            else { *px = make_float4(pix.x, pix.y, pix.z, taken); mode = M_PATH; }
                        }
                        if (finished) write_radiance(L, gx, slot >> 16, pix);
                        if constexpr (DIAG) {
                            if (L.tile_cost) {      // a measuring launch: what this path cost goes to its tile
                                const int row = fast_div((int)gx, S.dv_w), col = (int)gx - row * L.tile.tex_w;
                                atomicAdd(L.tile_cost + ((row >> 3) * S.tiles_x + (col >> 3)), lane_cost + 10u * ((slot & 0xffffu) + 1u));
                            }
                            lane_cost = 0u;
                        }
                    }
                } else {
                    need_segment = true;
                }
            }
        }
#ifdef PT_DIAG_SERVICE
        if (wave_times) st_switch(5);
#endif
        PT_SERVICE_PART();
        // refill: free lanes take new jobs
        uint32_t job_px = 0u, job_py = 0u;          // the pixel a lane has just been given (used below, in this step)
        if (do_b) {
            unsigned long long dead = __ballot(mode == M_DEAD && !need_segment);
            while (dead != 0ull && !feed_empty) {
                if (cur_used >= 64) {
                    cur_tile = have_at;
                    if (cur_tile >= ntiles) {
                        feed_empty = true;
                        if (wave_times) {
                            t_empty_rt = __builtin_amdgcn_s_memrealtime();
                            st_live_at_empty = (uint32_t)__popcll(__ballot(mode != M_DEAD || need_segment));
                        }
                        break;
                    }
                    if (L.job_reverse) cur_tile = ntiles - 1 - cur_tile;      // the launch's jobs in reverse order: bottom band first, top band last
                    if (grp_jobs == 0) {            // frame-major
                        cur_fslot = fast_div(cur_tile, dv_frame);
                        cur_ftile = cur_tile - cur_fslot * ntiles_frame;
                    } else {                        // groups of L.job_group tiles, every frame of a group before the next group
                        int g = fast_div(cur_tile, dv_grp), r = cur_tile - g * grp_jobs, gs = L.job_group;
                        const bool last = g >= grp_full;
                        if (last) { g = grp_full; r = cur_tile - grp_full * grp_jobs; gs = grp_last; }
                        cur_fslot = last ? fast_div(r, dv_last) : fast_div(r, dv_gs);
                        cur_ftile = g * L.job_group + (r - cur_fslot * gs);
                    }
                    if (L.tile_perm) cur_ftile = (int)L.tile_perm[cur_ftile];      // cost order (a scalar load: the position is wave-uniform)
                    have_at++;
                    if (--have == 0) draw(S);
                    cur_used = 0;
                }
                const int take = min((int)__popcll(dead), 64 - cur_used);
                const int rank = lane_rank(dead);
                const bool mine = ((dead >> lane) & 1ull) != 0ull && rank < take;
                bool got_job = false;
                if (mine) {
                    const int j = cur_used + rank;
                    const int fslot = cur_fslot, ftile = cur_ftile;
                    const int trow = fast_div(ftile, dv_tx);
                    const int px = (ftile - trow * tiles_x) * 8 + (j & 7);
                    const int ply = trow * 8 + (j >> 3);
                    const int pgy = local_to_global_row(ply, L.tile);
                    const bool ok = px < L.tile.tex_w && ply < L.tile.local_rows && pgy < L.tile.tex_h &&
                                    (uint32_t)px < res_w && (uint32_t)pgy < res_h;     // :425-427
                    if (ok) {
                        got_job = true;
                        seed = ((uint32_t)px + (uint32_t)pgy * res_w) + (un.frame + (uint32_t)fslot) * 719393u + PT_SEED;    // :435-436
                        // the pixel's coordinates are only needed for the camera ray, formed in this very step;
                        // what a lane carries through its walks: texel index, frame slot << 16 | bounce
                        job_px = (uint32_t)px; job_py = (uint32_t)pgy;
                        gx = (uint32_t)ply * (uint32_t)L.tile.tex_w + (uint32_t)px;
                        slot = (uint32_t)fslot << 16;
                        if (un.samples_per_frame != 1)      // (the pixel's running sum and sample count: see the path end above)
                            L.radiance[(size_t)fslot * L.slot_pixels + gx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        mode = M_PATH;
                    }
                }
                cur_used += take;
                u_pix += (uint32_t)__popcll(__ballot(got_job));
                // lanes that took a job (valid or not) leave the dead set; invalid ones rejoin next round
                dead = __ballot(mode == M_DEAD && !need_segment && !mine);
            }
        }
        PT_SERVICE_PART();
        if (count_on) st_path_lanes += (uint32_t)__popcll(__ballot(do_b && mode == M_PATH));
        if (do_b && mode == M_PATH) {
            // raytrace.wgsl:441-455: next sample's camera path, or the pixel is finished
            mode = M_DEAD;
            {
                float uvx, uvy;
                uint32_t pxx = job_px, pyy = job_py;
                {
                    if (un.samples_per_frame != 1) {
                        // a later sample of a multi-sample frame: the lane only carries the texel index; its pixel, once more
                        const int row = fast_div((int)gx, S.dv_w);
                        pxx = gx - (uint32_t)row * (uint32_t)L.tile.tex_w;
                        pyy = (uint32_t)local_to_global_row(row, L.tile);
                    }
                    // no sample to take (samplesPerFrame < 1: the shader's loop does not run and the sum, 0, is divided by
                    // f32(samplesPerFrame), :441-455) or no segment to trace (maxBounces < 1: every path returns 0, :376-377):
                    // the pixel is 0 / f32(samplesPerFrame)
                    // (the launch-uniform test sits BEHIND the camera ray below, which such a launch forms for nothing -- its lanes are dead
                    // afterwards and what it leaves in them is never read: a branch around the ray makes the compiler form o, d, the slot word,
                    // light and throughput in temporaries and copy them home where the two paths meet, 22 moves per step in EVERY launch)
                }
                // the pixel-only part -- uv, cameraToRay's direction, cam_pos + dir0 * focalDistance -- comes ready-made where the
                // context has formed it for this camera (L.cam_base: k_camera_base, the same operations on the same values); the
                // load is issued here, ahead of the disk sample below, whose ~70 instructions hide it
                f3 fbase;
                if (L.cam_base) {
                    const float4 cb = L.cam_base[gx];
                    fbase = xyz(cb);
                } else {
                    if (ASSUME || res_ordinary_ != 0) { uvx = div_pre((float)pxx, un.res_x, inv_res_x); uvy = div_pre((float)pyy, un.res_y, inv_res_y); }
                    else { uvx = (float)pxx / un.res_x; uvy = (float)pyy / un.res_y; }
                    const f3 dir0 = camera_direction(cf, un.aspect, uvx, uvy);
                    fbase = cam_pos + dir0 * un.focal_distance;
                }
                float jx, jy, kx, ky;
                rand_point_in_circle(seed, jx, jy);
                const f3 jitter = F3(jx * inv_res_x, jy * inv_res_y, 0.0f);
                const f3 focal = fbase + jitter;
                if (pinhole) {
                    // aperture == 0 (the reference's default, scene.ts:9): the lens offset is (+-0, +-0, 0) whatever the
                    // disk sample is (finite * 0), and cam_pos + (+-0) == cam_pos bit for bit unless a coordinate of
                    // cam_pos is -0 (excluded): what remains of randPointInCircle (raytrace.wgsl:283-287) is its two
                    // rand() calls, for the seed
                    (void)rand1(seed);
                    (void)rand1(seed);
                    o = cam_pos;
                } else {
                    rand_point_in_circle(seed, kx, ky);
                    const f3 jitter2 = F3(kx * un.aperture, ky * un.aperture, 0.0f);
                    o = cam_pos + jitter2;
                }
                d = normalize(focal - o);
                if (un.samples_per_frame < 1 || un.max_bounces < 1) {
                    const float z = un.samples_per_frame == 1 ? 0.0f : 0.0f / spf_f;
                    write_radiance(L, gx, slot >> 16, F3(z, z, z));
                } else {
                    need_segment = true;
                }
                slot &= 0xffff0000u;
                light = F3(0.0f, 0.0f, 0.0f);
                ray_color = F3(1.0f, 1.0f, 1.0f);
            }
        }
#ifdef PT_DIAG_SERVICE
        if (wave_times) st_switch(6);
#endif
        PT_SERVICE_PART();
        if (count_on) st_segment_lanes += (uint32_t)__popcll(__ballot(need_segment));
        {
            const uint32_t nseg = (uint32_t)__popcll(__ballot(need_segment));
            if (DIAG) u_rays += nseg;          // (shipped kernels: every segment ends in a hit or a miss, counted there)
            if (ASSUME || sc.nnodes != 0) u_box += nseg;      // the root box test
        }
        bool slow_segment = false;
        if (need_segment) {
            if constexpr (PARKG) {
                parkg[0] = ray_color.x; parkg[64] = ray_color.y; parkg[128] = ray_color.z;
                parkg[192] = light.x; parkg[256] = light.y; parkg[320] = light.z;
            } else {
                park[0] = ray_color.x; park[64] = ray_color.y; park[128] = ray_color.z;
                park[192] = light.x; park[256] = light.y; park[320] = light.z;
            }
        }
        ray_color = F3(0.0f, 0.0f, 0.0f);       // (dead until the lane's next shading: nothing to keep in registers -- zeros, which cost
        light = F3(0.0f, 0.0f, 0.0f);           // six moves per service step where they meet the values read back from the park, can be made
                                                // anew; an unset value would have to live in six registers through the whole walk: it spills)
        if (need_segment) {
            // raySceneIntersect + the root test of rayBVHIntersect, raytrace.wgsl:155-164, 205-211
            best.t = PT_INF; best.u = 0.0f; best.v = 0.0f; best.tri = -1;
            
            mode = M_SHADE;
            // A ray with a NaN in its origin or direction (normalize() of a zero vector upstream: it happens once in
            // ~10^8 paths on real meshes) hits nothing in the reference -- Moller-Trumbore's determinant or its u, v, t
            // are NaN for EVERY triangle and every acceptance test is then false (raytrace.wgsl:78-116) -- but passes
            // every box test (min / max drop the NaNs, :136-143), i.e. walks the WHOLE tree: 18 s for one such ray on
            // the 10 M-triangle forest.  The culling walks, which do not reproduce the reference's test counts anyway,
            // take the known answer: a miss.  (Variants 1-8 walk it.)
            const bool nan_ray = CULL && (!(d.x == d.x) || !(d.y == d.y) || !(d.z == d.z) ||
                                          !(o.x == o.x) || !(o.y == o.y) || !(o.z == o.z));
            if ((ASSUME || sc.nnodes != 0) && !nan_ray) {
                pre = ray_prepare(o, d, sc.flags);
                slow_segment = (pre.flags & 8u) != 0u;
                if (CULL) cull_setup(d, pre, sc.cull_ka, sc.cull_kb, cull_ka, cull_kb);
                
                // (WIDE, root box nested: a ray that the reference's root test (raytrace.wgsl:160-164) rejects is rejected by every box of
                // the root packet too -- the fp32 slab test is monotone under nesting, the argument of the wide collapse -- so the test
                // is left to the first node step: bounce rays start inside the scene's box and pass it practically always, and its
                // ~45 exact-slab instructions were paid by every segment.  Measured +1.5 % / +1.9 %.  Going further -- the root PACKET's
                // four box tests in the segment start, from scalars of the service block: one node step less per segment, but 230
                // more instructions in the issue-bound service step -- lost 6 %: profiles/r03_i_skip_root_ab.log)
                const bool skip_root = WIDE && (sc.flags & 2u) != 0u;
                if (skip_root || ray_aabb_pre(o, d, pre, !ASSUME && (sc.flags & 1u) == 0u, S.root_mn[0], S.root_mn[1], S.root_mn[2], S.root_mx[0], S.root_mx[1], S.root_mx[2])) {
                    if (!ASSUME && DEFER && (sc.root_ref & PT_REF_LEAF)) {      // one-triangle scene
                        stack[(DEPTH - 1) * 64] = sc.root_ref & 0x7fffffffu;
                        sp = 0; nl = 1;
                    } else if constexpr (W8) {
                        // the octant of the ray's direction signs (bits 4-6 of the ray's flags: set where the component is not negative), and a
                        // first entry that stands for the root packet alone: child base 0, one hit, no internal-slot mask (rank 0 whatever the slot)
                        pre.flags |= (pre.ix < 0.0f ? 0u : 16u) | (pre.iy < 0.0f ? 0u : 32u) | (pre.iz < 0.0f ? 0u : 64u);
                        stack[0] = 0u; stack[64] = 1u;
                        sp = 1; nl = 0;
                    } else {
                        st_store(0, WIDE ? sc.wide_root : sc.root_ref);
                        sp = 1; nl = 0;
                    }
                    mode = M_TRAV;
                }
            }
        }
        if (DIAG) u_slow += (uint32_t)__popcll(__ballot(slow_segment));
#undef PT_SERVICE_PART
    }

    if constexpr (LITE && !DIAG) {
        if (L.wave_times && lane == 0) {       // (the same slots as the diagnostic twin's below; what it does not measure stays 0)
            uint64_t *w = L.wave_times + (size_t)blockIdx.x * 16;
            w[8] = st_tri_steps;
            w[9] = 0; w[10] = 0; w[11] = 0;
            w[12] = ((uint64_t)st_hit_steps << 32) | st_b_steps;
            w[13] = 0; w[14] = 0; w[15] = 0;
            w[0] = t_begin_rt;
            w[1] = 0;
            w[2] = __builtin_amdgcn_s_memrealtime();
            w[3] = __builtin_amdgcn_s_memtime() - t_begin_clk;
            w[4] = ((uint64_t)st_walk_steps << 32) | st_walk_lanes;
            w[5] = ((uint64_t)st_service_steps << 32) | st_leaf_lanes;
            w[6] = ((uint64_t)st_shade_lanes << 32) | st_hit_lanes;
            w[7] = ((uint64_t)st_path_lanes << 32) | st_segment_lanes;
        }
    }
    if (wave_times && lane == 0) {
        uint64_t *w = wave_times + (size_t)blockIdx.x * 16;
        st_switch(2);
        w[8] = ((uint64_t)st_parked << 32) | st_tri_steps;      // (high half: triangles parked in the wave, summed over its triangle steps)
        w[9] = st_cyc_node; w[10] = st_cyc_tri; w[11] = st_cyc_service;
        w[12] = ((uint64_t)st_hit_steps << 32) | st_b_steps;
#ifdef PT_DIAG_SERVICE
        w[12] = st_cyc_part[0]; w[13] = st_cyc_part[1]; w[14] = st_cyc_part[2]; w[15] = st_cyc_part[3];
#else
        w[13] = ((uint64_t)st_tail_node << 32) | st_tail_tri;
        w[14] = ((uint64_t)st_tail_service << 32) | st_live_at_empty;
        w[15] = st_tail_lanes;
#endif
        w[0] = t_begin_rt;
        w[1] = t_empty_rt;
        w[2] = __builtin_amdgcn_s_memrealtime();
        w[3] = __builtin_amdgcn_s_memtime() - t_begin_clk;
        w[4] = ((uint64_t)st_walk_steps << 32) | st_walk_lanes;
        w[5] = ((uint64_t)st_service_steps << 32) | st_leaf_lanes;
        w[6] = ((uint64_t)st_shade_lanes << 32) | st_hit_lanes;
        w[7] = ((uint64_t)st_path_lanes << 32) | st_segment_lanes;
    }
    // Self-cleaning work queue: the last wave to leave resets the head and the exit counter, so
    // launches need no memset in front of them (a memset kernel would have to wait for a free
    // slot among the previous batch's persistent waves).
    if (lane == 0) {
        const uint32_t left = atomicAdd(L.tile_counter + 1, 1u);
        if (left == gridDim.x - 1u) {
            __hip_atomic_store(L.tile_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(L.tile_counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const uint32_t s_rays = DIAG ? u_rays : u_hit + u_miss, s_box = wave_sum(cnt.box) + u_box, s_tri = wave_sum(cnt.tri) + u_tri;
    const uint32_t s_hit = u_hit, s_miss = u_miss;
    const uint32_t s_ovf = wave_sum(cnt.overflow), s_pix = u_pix, s_slow = u_slow;
    if (lane == 0 && L.block_counters) {
        uint64_t *c = L.block_counters + (size_t)blockIdx.x * CNT_COUNT;
        c[CNT_RAYS] += s_rays; c[CNT_BOX] += s_box; c[CNT_TRI] += s_tri; c[CNT_HIT] += s_hit;
        c[CNT_MISS] += s_miss; c[CNT_OVERFLOW] += s_ovf; c[CNT_PIXELS] += s_pix;
        c[CNT_RESERVED] += s_slow;
    }
}

int raytrace_persistent_blocks(const Tile &tile, int nframes, int waves_per_cu, int num_cus, bool tuned, int waves_per_simd)
{
    const int ntiles = raytrace_grid_blocks(tile);
    if (num_cus <= 0) num_cus = 256;
    if (waves_per_simd <= 0) waves_per_simd = tuned ? SM_TUNED_WAVES_PER_SIMD : SM_OTHER_WAVES_PER_SIMD;
    if (waves_per_cu > 4 * waves_per_simd) waves_per_cu = 0;          // (more than the build can keep resident: its own width)
    if (waves_per_cu <= 0 || waves_per_cu > PT_MAX_WAVES_PER_CU) {
        waves_per_cu = 4 * waves_per_simd;
        // A small launch (an interactive host: one or two frames, or a small image) gets fewer waves: at least 8 jobs each,
        // at least 4 per CU.  With a handful of jobs per wave a launch is all ramp and drain, and a launch that fills every
        // wave slot keeps its successor out until its own waves exit; narrower launches overlap.  One 1080p frame per
        // launch: 16 waves per CU, -6 % time; 640 x 360: 4, -20 % (profiles/r03_k_small_launches.log; fewer still is
        // faster on the default scene and slower on every other one tried: MI3PT_OPT_WAVES_PER_CU)
        // Up to ~3 frames of 1080p per launch, 16 per CU beats 20 on every scene tried (-5 ... -12 %: the successor finds
        // four wave slots per CU free from the start); from 4 frames on the full width wins.
        const long long jobs = (long long)ntiles * (nframes > 0 ? nframes : 1);
        const long long want = (jobs + (long long)num_cus * 8 - 1) / ((long long)num_cus * 8);
        if (want < 16) waves_per_cu = want < 4 ? 4 : (int)want;
        else if (want < 48 && waves_per_cu > 16) waves_per_cu = 16;
    }
    int resident = num_cus * waves_per_cu;                            // the device's own CU count (hipDeviceProp_t)
    if (resident > PT_MAX_RESIDENT_WAVES) resident = PT_MAX_RESIDENT_WAVES;
    // One wave per JOB at most -- (frame slot, tile) pairs, not tiles of one frame.  Until round 4 this read `ntiles`: a rank of an
    // 8-way split of 1080p has 17 x 240 = 4 080 tiles per frame, so its 512-frame launches ran 4 080 of the 5 120 wave slots
    // (16 instead of 20 per CU) -- the "unexplained +14 % per pixel" of a rank of eight (profiles/r04_a_rank_penalty.log:
    // fewer wave cycles per ray than the whole image, yet more time per ray).
    const long long jobs = (long long)ntiles * (nframes > 0 ? nframes : 1);
    return jobs < (long long)resident ? (int)jobs : resident;
}

int raytrace_grid_blocks(const Tile &tile)
{
    const int tiles_x = (tile.tex_w + 7) / 8;
    const int tiles_y = (tile.local_rows + 7) / 8;
    return tiles_x * tiles_y;
}

// Frame slot and bounce share a word in the state-machine kernel, and the sample count of a multi-sample frame is a float in the
// pixel's texel: launches beyond these (absurd) limits run the per-pixel kernel, frame by frame.
static bool launch_packs(const RtLaunch &L)
{
    return L.un.max_bounces < 65536 && L.nframes <= 65535 && L.un.samples_per_frame <= 16777216;
}
// The lean build (DIAG = false: no step statistics, the step-voting options as constants, the service step's scalars in
// memory) runs unless a diagnostic buffer is bound or a step-voting option was changed (mi3pt_debug_set_option).
static bool launch_is_lean(const RtLaunch &L)
{
    return (!L.wave_times || L.diag_lite) && !L.tile_cost && (L.walk_min == PT_DEFAULT_WALK_MIN || L.walk_min == PT_DEEP_WALK_MIN) && L.leaf_min == PT_DEFAULT_LEAF_MIN && L.shade_split == PT_DEFAULT_SHADE_SPLIT &&
           L.tail_policy == PT_DEFAULT_TAIL_POLICY && L.job_chunk == PT_DEFAULT_JOB_CHUNK && L.tri_pair == 1 && L.service != nullptr;
}
// What the lean builds of the culling walks have as constants (ASSUME in the kernel): a scene with nodes whose root is an
// internal node with a guard-range box, and a resolution of ordinary magnitude.  Every tree the culling walks are offered for
// has nodes and an internal root; the rest fails for scenes ~1e18 across or images ~1e12 wide.
static bool launch_assumptions_hold(const RtLaunch &L)
{
    return L.scene.nnodes != 0 && (L.scene.flags & 1u) != 0u && (L.scene.root_ref & PT_REF_LEAF) == 0u &&
           L.un.res_x >= 9.5367431640625e-07f && L.un.res_x <= 1.099511627776e12f &&
           L.un.res_y >= 9.5367431640625e-07f && L.un.res_y <= 1.099511627776e12f;
}

// The kernel a launch runs for a requested variant.  kind: 0 = per-pixel kernel (variant 1 / 2), 1 = state-machine kernel
// (variant 4, 7, 9 .. 12; experiment builds: 5, 6, 8), 2 = k_raytrace_persistent (variant 3, experiment builds).
static RtRoute route_launch(const RtLaunch &L, int variant)
{
    RtRoute r = { 0, variant <= 1 ? 1 : 2, false, 0 };
    if (variant < 3) return r;
#ifdef MI3PT_EXPERIMENTS
    if (variant == 3) { r.kind = 2; r.variant = 3; return r; }
#else
    if (variant == 3) variant = 4;                       // (release builds: the experiment variants resolve to their nearest walk)
    if (variant == 5 || variant == 6) variant = 4;
    if (variant == 8) variant = 7;
#endif
    if (!launch_packs(L)) return r;                      // per-pixel kernel 2, frame by frame
    r.kind = 1;
    r.lean = launch_is_lean(L);
    if (variant == 14 && !(L.scene.cw8 && L.scene.tripk8 && (L.scene.flags & 2u))) variant = 13;      // (the 8-wide walk never tests the root's own box)
    if (variant == 13 && !(L.scene.cwide && L.scene.tripk64)) variant = 10;
    if (L.walk_min == PT_DEEP_WALK_MIN && variant != 13 && variant != 14) r.lean = false;        // (the deep-walk threshold is instantiated for the compressed-wide walk only)
#ifndef MI3PT_EXPERIMENTS
    if (variant == 11 || variant == 12) variant = 10;       // (release builds: superseded by 13; the exact-packet walk with the exact slab test stands in -- same bits)
#endif
    if (variant >= 9 && variant <= 14 && r.lean && !launch_assumptions_hold(L)) variant = L.scene.leaf_cap >= 4 ? 7 : 4;
    if (variant >= 10 && variant <= 14 && !r.lean) variant = 10;      // the diagnostic twin of the wide walks runs the exact slab test: same bits
#ifndef MI3PT_EXPERIMENTS
    if (variant < 9 && !r.lean) {
        // release builds carry diagnostic twins for the culling walks only: the lean build runs (options and the diagnostic buffer
        // are ignored) -- or, without a service block, the per-pixel kernel
        if (!L.service) { r.kind = 0; r.variant = 2; return r; }
        r.lean = true;
    }
#endif
    r.variant = variant;
    return r;
}
bool raytrace_variant_fuses(int variant)
{
#ifdef MI3PT_EXPERIMENTS
    return variant <= 3;
#else
    return variant <= 2;
#endif
}
// Waves per SIMD of the build a route runs.  The compressed-wide walk has two lean builds per instantiation: FIVE waves (96 registers, a
// 24-entry LDS stack) and SIX (80 registers with a handful spilled around the walk loop; 19 entries -- or 25 with the parked path state in
// memory, for very large trees).  Six waves overlap more of the per-step round trips: +2 .. +5 % on launches that run for tens of
// milliseconds.  But a sixth more paths are in flight when the job queue runs empty, and that drain is paid once per launch: a rank of an
// 8-way split (a 10 ms launch) is 2 .. 3 % SLOWER with six (profiles/r05_i_six_waves_by_launch_size.log).  So the choice goes by the size of
// the launch: jobs (tiles x frames) per resident wave.  L.six_waves forces it (MI3PT_OPT_SIX_WAVES).  Other lean builds: five; twins: four.
#ifndef PT_SIX_WAVES_MIN_JOBS
#define PT_SIX_WAVES_MIN_JOBS 1500000      // (round 6, re-swept on the cheaper node step: six waves win from ~1.3 M jobs on -- profiles/r06_i_six_waves_threshold.log; was 2.5 M)
#endif
static int route_waves_per_simd(const RtLaunch &L, const RtRoute &r)
{
    if (!(r.kind == 1 && r.lean)) return SM_OTHER_WAVES_PER_SIMD;
    if (r.variant == 13 || r.variant == 14) {
#ifdef MI3PT_EXPERIMENTS
        if (L.wave_times && L.diag_lite) return SM_TUNED_WAVES_PER_SIMD;      // (the lean build + lane counts: five)
#endif
        if (L.six_waves >= 0) return L.six_waves ? SM_SIX_WAVES_PER_SIMD : SM_TUNED_WAVES_PER_SIMD;
        const long long jobs = (long long)raytrace_grid_blocks(L.tile) * (L.nframes > 0 ? L.nframes : 1);
        // (a job of a very large tree -- the walk_min-44 builds -- is an order of magnitude longer)
        return jobs >= (L.walk_min == PT_DEEP_WALK_MIN ? PT_SIX_WAVES_MIN_JOBS / 10 : PT_SIX_WAVES_MIN_JOBS) ? SM_SIX_WAVES_PER_SIMD : SM_TUNED_WAVES_PER_SIMD;
    }
    return SM_TUNED_WAVES_PER_SIMD;
}
static int persistent_blocks_for(const RtLaunch &L, const RtRoute &r)
{
    return raytrace_persistent_blocks(L.tile, L.nframes, L.waves_per_cu, L.num_cus, r.kind == 1 && r.lean, route_waves_per_simd(L, r));
}
RtRoute raytrace_route(const RtLaunch &L, int variant)
{
    RtRoute r = route_launch(L, variant);
    r.blocks = r.kind == 1 ? persistent_blocks_for(L, r) : raytrace_grid_blocks(L.tile);
    r.waves = r.kind == 1 ? route_waves_per_simd(L, r) : 0;
    r.ymax = r.kind == 1 && r.lean && r.variant >= 12 && (L.scene.flags & 4u) != 0u;
    r.walk_min = r.kind == 1 ? (r.lean ? ((r.variant == 13 || r.variant == 14) && L.walk_min == PT_DEEP_WALK_MIN ? PT_DEEP_WALK_MIN : PT_DEFAULT_WALK_MIN) : L.walk_min) : 0;
    return r;
}

// The lean builds read their service step's scalars from L.service: filled here, in stream order, by one wave.  Its own
// call so that the caller can put its timing event between this and the raytrace kernel (the little kernel waits for the
// first wave slot a draining predecessor frees; that wait is not the raytrace kernel's time).
void launch_raytrace_setup(const RtLaunch &L, bool fuse, int variant, hipStream_t s)
{
    (void)fuse;
    const RtRoute r = route_launch(L, variant);
    if (raytrace_grid_blocks(L.tile) > 0 && r.kind == 1 && r.lean)
        hipLaunchKernelGGL(k_rt_service_setup, dim3(1), dim3(64), 0, s, L, L.service);
}

// (the caller has called launch_raytrace_setup with the same arguments on the same stream; `fuse` only where
// raytrace_variant_fuses(variant))
void launch_raytrace(const RtLaunch &L, bool fuse, int variant, hipStream_t s)
{
    const int blocks = raytrace_grid_blocks(L.tile);
    if (blocks <= 0) return;
    const dim3 block(64);
    const RtRoute r = route_launch(L, variant);
    if (r.kind == 1) {
        const dim3 grid(persistent_blocks_for(L, r));
#define PT_SM(...) hipLaunchKernelGGL((k_raytrace_sm<__VA_ARGS__>), grid, block, 0, s, L)
        //                               DEFER  CULL   WIDE   FILT   YMAX   DIAG
#ifdef MI3PT_EXPERIMENTS
        if (r.lean && L.wave_times && L.diag_lite && r.variant >= 10) switch (r.variant) {      // the lean build + lane counts
            case 14: if (L.scene.flags & 4u) PT_SM(true, true, true, true, true, false, false, true, true, PT_DEFAULT_WALK_MIN, SM_TUNED_WAVES_PER_SIMD, true);
                     else PT_SM(true, true, true, true, false, false, false, true, true, PT_DEFAULT_WALK_MIN, SM_TUNED_WAVES_PER_SIMD, true); break;
            case 13: if (L.scene.flags & 4u) PT_SM(true, true, true, true, true, false, false, true, true); else PT_SM(true, true, true, true, false, false, false, true, true); break;
            case 12: PT_SM(true,  true,  true,  true,  true,  false, false, true); break;
            case 11: PT_SM(true,  true,  true,  true,  false, false, false, true); break;
            default: PT_SM(true,  true,  true,  false, false, false, false, true); break;
        } else
#endif
#ifndef PT_X_TOP_CW
#define PT_TOP_CW false
#else
#define PT_TOP_CW true          // (A/B build: the first PT_X_TOP_CW compressed packets staged in LDS per wave)
#endif
        if (r.lean && r.variant == 14) {        // eight-wide compressed packets
            const bool six = route_waves_per_simd(L, r) == SM_SIX_WAVES_PER_SIMD, ymax = (L.scene.flags & 4u) != 0u;
            if (L.walk_min == PT_DEEP_WALK_MIN) {
                if (six) { if (ymax) PT_SM(true, true, true, true, true, false, false, false, true, PT_DEEP_WALK_MIN, SM_SIX_WAVES_PER_SIMD, true); else PT_SM(true, true, true, true, false, false, false, false, true, PT_DEEP_WALK_MIN, SM_SIX_WAVES_PER_SIMD, true); }
                else { if (ymax) PT_SM(true, true, true, true, true, false, false, false, true, PT_DEEP_WALK_MIN, SM_TUNED_WAVES_PER_SIMD, true); else PT_SM(true, true, true, true, false, false, false, false, true, PT_DEEP_WALK_MIN, SM_TUNED_WAVES_PER_SIMD, true); }
            } else {
                if (six) { if (ymax) PT_SM(true, true, true, true, true, false, false, false, true, PT_DEFAULT_WALK_MIN, SM_SIX_WAVES_PER_SIMD, true); else PT_SM(true, true, true, true, false, false, false, false, true, PT_DEFAULT_WALK_MIN, SM_SIX_WAVES_PER_SIMD, true); }
                else { if (ymax) PT_SM(true, true, true, true, true, false, false, false, true, PT_DEFAULT_WALK_MIN, SM_TUNED_WAVES_PER_SIMD, true); else PT_SM(true, true, true, true, false, false, false, false, true, PT_DEFAULT_WALK_MIN, SM_TUNED_WAVES_PER_SIMD, true); }
            }
        } else
        if (r.lean && r.variant == 13) {        // compressed wide packets (SceneRefs::flags bit 2: the one-axis culling condition suits this scene)
            const bool six = route_waves_per_simd(L, r) == SM_SIX_WAVES_PER_SIMD, ymax = (L.scene.flags & 4u) != 0u;
            if (L.walk_min == PT_DEEP_WALK_MIN) {
                if (six) { if (ymax) PT_SM(true, true, true, true, true, false, PT_TOP_CW, false, true, PT_DEEP_WALK_MIN, SM_SIX_WAVES_PER_SIMD); else PT_SM(true, true, true, true, false, false, PT_TOP_CW, false, true, PT_DEEP_WALK_MIN, SM_SIX_WAVES_PER_SIMD); }
                else { if (ymax) PT_SM(true, true, true, true, true, false, PT_TOP_CW, false, true, PT_DEEP_WALK_MIN); else PT_SM(true, true, true, true, false, false, PT_TOP_CW, false, true, PT_DEEP_WALK_MIN); }
            } else {
                if (six) { if (ymax) PT_SM(true, true, true, true, true, false, PT_TOP_CW, false, true, PT_DEFAULT_WALK_MIN, SM_SIX_WAVES_PER_SIMD); else PT_SM(true, true, true, true, false, false, PT_TOP_CW, false, true, PT_DEFAULT_WALK_MIN, SM_SIX_WAVES_PER_SIMD); }
                else { if (ymax) PT_SM(true, true, true, true, true, false, PT_TOP_CW, false, true); else PT_SM(true, true, true, true, false, false, PT_TOP_CW, false, true); }
            }
        } else
        if (r.lean) switch (r.variant) {
#ifdef MI3PT_EXPERIMENTS
            case 12: PT_SM(true,  true,  true,  true,  true,  false); break;
            case 11: PT_SM(true,  true,  true,  true,  false, false); break;
#endif
            case 10: PT_SM(true,  true,  true,  false, false, false); break;
            case 9:  PT_SM(true,  true,  false, false, false, false); break;
            case 7:  PT_SM(true,  false, false, false, false, false); break;
            default: PT_SM(false, false, false, false, false, false); break;
        } else switch (r.variant) {
            case 10: PT_SM(true,  true,  true,  false, false, true); break;
            case 9:  PT_SM(true,  true,  false, false, false, true); break;
#ifdef MI3PT_EXPERIMENTS
            case 8:  PT_SM(true,  false, false, false, false, true, true); break;
            case 7:  PT_SM(true,  false, false, false, false, true); break;
            case 6:  PT_SM(false, false, false, false, false, true, true); break;
            default: PT_SM(false, false, false, false, false, true); break;
#else
            default: break;          // (not reached: route_launch)
#endif
        }
#undef PT_SM
        return;
    }
#ifdef MI3PT_EXPERIMENTS
    if (r.kind == 2) {
        const dim3 grid(raytrace_persistent_blocks(L.tile, 1, L.waves_per_cu, L.num_cus, false));
        if (fuse) hipLaunchKernelGGL((k_raytrace_persistent<true>), grid, block, 0, s, L);
        else hipLaunchKernelGGL((k_raytrace_persistent<false>), grid, block, 0, s, L);
        return;
    }
#endif
    // per-pixel kernels: one launch per frame of the batch
    const dim3 grid(blocks);
    for (int k = 0; k < (L.nframes > 0 ? L.nframes : 1); k++) {
        RtLaunch F = L;
        F.nframes = 1;
        F.un.frame = L.un.frame + (uint32_t)k;
        F.acc.frame = L.acc.frame + (uint32_t)k;
        F.radiance = L.radiance + (size_t)k * L.slot_pixels;
        if (r.variant == 2) {
            if (fuse) hipLaunchKernelGGL((k_raytrace<true, 2>), grid, block, 0, s, F);
            else hipLaunchKernelGGL((k_raytrace<false, 2>), grid, block, 0, s, F);
        } else {
            if (fuse) hipLaunchKernelGGL((k_raytrace<true, 1>), grid, block, 0, s, F);
            else hipLaunchKernelGGL((k_raytrace<false, 1>), grid, block, 0, s, F);
        }
    }
}

// ---------------------------------------------------------------------------------
// accumulate.wgsl:12-29 as its own pass
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_accumulate(const AccUniforms acc, const Tile tile,
                                                    const float4 *__restrict__ input,
                                                    float4 *__restrict__ accum, int store_f16)
{
    const size_t n = (size_t)tile.local_rows * tile.tex_w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ly = (int)(i / tile.tex_w);
        const int gx = (int)(i - (size_t)ly * tile.tex_w);
        const int gy = local_to_global_row(ly, tile);
        if ((uint32_t)gx >= acc.res_w || (uint32_t)gy >= acc.res_h) continue;
        const float4 c = input[i];
        const float4 p = accum[i];
        const f3 nc = accumulate_texel(acc, xyz(c), xyz(p));
        accum[i] = make_float4(store_round(nc.x, store_f16), store_round(nc.y, store_f16),
                               store_round(nc.z, store_f16), 1.0f);
    }
}

// The running mean over `nframes` consecutive frames' radiance slots, applied per pixel in
// frame order (frame, frame+1, ...) -- the same sequence of accumulate.wgsl passes, with the
// accumulator read and written once.
__global__ void __launch_bounds__(256) k_accumulate_batch(const AccUniforms acc0, const Tile tile,
                                                          const float4 *__restrict__ slots, size_t slot_pixels,
                                                          int nframes, float4 *__restrict__ accum, int store_f16)
{
    const size_t n = (size_t)tile.local_rows * tile.tex_w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ly = (int)(i / tile.tex_w);
        const int gx = (int)(i - (size_t)ly * tile.tex_w);
        const int gy = local_to_global_row(ly, tile);
        if ((uint32_t)gx >= acc0.res_w || (uint32_t)gy >= acc0.res_h) continue;
        f3 p = xyz(accum[i]);
        for (int k = 0; k < nframes; k++) {
            AccUniforms a = acc0;
            a.frame = acc0.frame + (uint32_t)k;
            const f3 nc = accumulate_texel(a, xyz(slots[(size_t)k * slot_pixels + i]), p);
            p = F3(store_round(nc.x, store_f16), store_round(nc.y, store_f16), store_round(nc.z, store_f16));
        }
        accum[i] = make_float4(p.x, p.y, p.z, 1.0f);
    }
}

void launch_accumulate_batch(const AccUniforms &acc0, const Tile &tile, const float4 *slots, size_t slot_pixels,
                             int nframes, float4 *accum, int store_f16, hipStream_t s)
{
    const size_t n = (size_t)tile.local_rows * tile.tex_w;
    if (n == 0 || nframes <= 0) return;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_accumulate_batch, dim3(blocks), dim3(256), 0, s, acc0, tile, slots, slot_pixels, nframes,
                       accum, store_f16);
}

void launch_accumulate(const AccUniforms &acc, const Tile &tile, const float4 *input, float4 *accum,
                       int store_f16, hipStream_t s)
{
    const size_t n = (size_t)tile.local_rows * tile.tex_w;
    if (n == 0) return;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_accumulate, dim3(blocks), dim3(256), 0, s, acc, tile, input, accum, store_f16);
}

// ---------------------------------------------------------------------------------
// fullscreen.wgsl
// ---------------------------------------------------------------------------------
// repeat addressing: v mod n.  Sample positions stay within one period of the texture, so the
// common case is a conditional add / subtract; the integer modulo is the general fallback.
PT_DEV int wrapi(int v, int n)
{
    if (v >= -n && v < 2 * n) return v < 0 ? v + n : (v >= n ? v - n : v);
    const int m = v % n;
    return m < 0 ? m + n : m;
}

// textureSample(inputTexture, sampler{linear, repeat}) -- fullscreen.ts:49-57
PT_DEV float4 sample_repeat(const float4 *tex, int W, int H, float u, float v)
{
    const float x = u * (float)W - 0.5f;
    const float y = v * (float)H - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    if (!(x0f > -1.0e9f && x0f < 1.0e9f)) x0f = 0.0f;
    if (!(y0f > -1.0e9f && y0f < 1.0e9f)) y0f = 0.0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int xa = wrapi(x0, W), xb = wrapi(x0 + 1, W);
    const int ya = wrapi(y0, H), yb = wrapi(y0 + 1, H);
    const float4 p00 = tex[(size_t)ya * W + xa];
    const float4 p10 = tex[(size_t)ya * W + xb];
    const float4 p01 = tex[(size_t)yb * W + xa];
    const float4 p11 = tex[(size_t)yb * W + xb];
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    float4 r;
    r.x = (p00.x * wx0 + p10.x * fx) * wy0 + (p01.x * wx0 + p11.x * fx) * fy;
    r.y = (p00.y * wx0 + p10.y * fx) * wy0 + (p01.y * wx0 + p11.y * fx) * fy;
    r.z = (p00.z * wx0 + p10.z * fx) * wy0 + (p01.z * wx0 + p11.z * fx) * fy;
    r.w = (p00.w * wx0 + p10.w * fx) * wy0 + (p01.w * wx0 + p11.w * fx) * fy;
    return r;
}

// fullscreen.wgsl:53-86 with sigma 5, kSigma 1, threshold 0.08 (:117-119).
//
// The loop bounds, the tap offsets d / size and the spatial weight
// exp(-dot(d,d) * invSigmaQx2) * invSigmaQx2PI depend only on the tap, not on the pixel: a setup kernel evaluates them
// once per change of the pass's uniforms (one column of taps per thread, with the same float loop `y = y + 1.0` and the
// same exp1) into a table in device memory, and every pixel reuses them -- one exp and no division per tap instead of
// two exps and two divisions, identical bits.  The table is read with scalar loads, a column (128 B) at a time: offsets
// and weights are instruction operands in scalar registers, the loop over a column's taps is unrolled.
#define PT_DN_COLS 11          // x = -radius .. radius for radius = round(kSigma * sigma) = 5
#define PT_DN_ROWS 12

struct DenoiseColumn {                     // 32 dwords
    float ox;                              // x / size.x
    int count;                             // taps in this column
    float oy[PT_DN_ROWS];                  // y / size.y
    float blur[PT_DN_ROWS];                // spatial weight
    float pad[6];
};
struct DenoiseTaps {
    DenoiseColumn col[PT_DN_COLS];
    float ox_lo, ox_hi, oy_lo, oy_hi;      // least / greatest offsets over all taps
    int finite;                            // are all offsets finite
    int pad[3];
};
static_assert(sizeof(DenoiseColumn) == 128 && sizeof(DenoiseTaps) == PT_DN_COLS * 128 + 32, "scalar-load layout");

size_t fullscreen_taps_bytes() { return sizeof(DenoiseTaps); }

__global__ void __launch_bounds__(64) k_fullscreen_setup(const FsUniforms fs, DenoiseTaps *__restrict__ taps)
{
    __shared__ float lo_s[PT_DN_COLS], hi_s[PT_DN_COLS];
    __shared__ int fin_s[PT_DN_COLS];
    const int column = (int)threadIdx.x;
    if (column < PT_DN_COLS) {
        const float sigma = 5.0f, k_sigma = 1.0f;
        const float INV_PI = 0.31830988618379067153776752674503f;
        const float radius = rintf(k_sigma * sigma);
        const float rad_q = radius * radius;
        const float inv_sigma_qx2 = 0.5f / (sigma * sigma);
        const float inv_sigma_qx2pi = INV_PI * inv_sigma_qx2;
        float x = -radius;
        for (int c = 0; c < column; c++) x = x + 1.0f;      // the reference's float loop variable
        DenoiseColumn &C = taps->col[column];
        int n = 0;
        float lo = 0.0f, hi = 0.0f;
        const float ox = x / fs.res_x;
        bool finite = fabsf(ox) < PT_INF;
        if (x <= radius) {
            const float pt = sqrtf(rad_q - x * x);
            for (float y = -pt; y <= pt && n < PT_DN_ROWS; y = y + 1.0f) {
                const float dd = x * x + y * y;
                const float oy = y / fs.res_y;
                C.oy[n] = oy;
                C.blur[n] = ptm::exp1(-dd * inv_sigma_qx2) * inv_sigma_qx2pi;
                finite = finite && fabsf(oy) < PT_INF;
                lo = n == 0 ? oy : fminf(lo, oy);
                hi = n == 0 ? oy : fmaxf(hi, oy);
                n++;
            }
        }
        for (int k = n; k < PT_DN_ROWS; k++) { C.oy[k] = 0.0f; C.blur[k] = 0.0f; }
        for (float &q : C.pad) q = 0.0f;
        C.ox = ox;
        C.count = n;
        lo_s[column] = lo; hi_s[column] = hi; fin_s[column] = finite ? 1 : 0;
    }
    __syncthreads();
    if (column == 0) {
        float ox_lo = taps->col[0].ox, ox_hi = ox_lo, oy_lo = lo_s[0], oy_hi = hi_s[0];
        int finite = fin_s[0];
        for (int c = 1; c < PT_DN_COLS; c++) {
            ox_lo = fminf(ox_lo, taps->col[c].ox);
            ox_hi = fmaxf(ox_hi, taps->col[c].ox);
            oy_lo = fminf(oy_lo, lo_s[c]);
            oy_hi = fmaxf(oy_hi, hi_s[c]);
            finite &= fin_s[c];                     // (fminf / fmaxf skip a NaN)
        }
        taps->ox_lo = ox_lo; taps->ox_hi = ox_hi; taps->oy_lo = oy_lo; taps->oy_hi = oy_hi;
        taps->finite = finite;
        taps->pad[0] = taps->pad[1] = taps->pad[2] = 0;
    }
}

// ---- the de-noise pass over an LDS copy of the block's texels --------------------------------------------------
// A 16 x 16 block of canvas pixels reads, tap by tap, the texels around it: 85 taps x 4 texels per pixel, every texel
// wanted by ~ 85 pixels.  The block copies the PT_FS_TW x PT_FS_TH texels around it into LDS once (repeat addressing
// applied by the copy).  A wave whose lanes' taps ALL lie inside the copy -- each lane checks the two extreme offsets
// per axis: floor((c + o) * n - 0.5) is monotone in o, every step being a monotone fp32 operation, so the taps between
// lie between -- reads from there; any other wave (a `scaling` > 1, resolution uniforms that are not the texture's,
// NaN) reads the texture as before.  Arithmetic and its order are sample_repeat's, per channel
//     (p00 * wx0 + p10 * fx) * wy0 + (p01 * wx0 + p11 * fx) * fy,
// so the bits are; the horizontal half (one "row value") of a tap's lower row is the next tap's upper row whenever the
// column's taps step by whole texels (the same expression of the same operands: evaluated once, reused).
// 16 pixels + 2 x (radius 5 + 1 for floor's side) = 28 texels, + 1 on either side: (px + 0.5) / w * w - 0.5 rounds to just
// below px for some px and to px for others (1920: every few pixels), so the texel pairs of a block's pixels start at
// px - 1 or at px, pixel by pixel, whatever the block's first pixel does
#define PT_FS_TW 30
#define PT_FS_TH 30
#define PT_FS_MARGIN 7

struct FsTile {
    const float4 *lds;       // PT_FS_TW x PT_FS_TH texels, row-major
    const float4 *tex;
    int x0, y0;              // texel (before wrapping) of the tile's first column / row
    int W, H;
};

// one axis of textureSample: texel index before wrapping, weights
struct FsAxis { int i0; float f, w0; };

PT_DEV float fs_texel(float coord, int n) { return floorf(coord * (float)n - 0.5f); }

// GUARD = false: the caller knows the texel index to be an ordinary number (it lies inside the tile)
template <bool GUARD>
PT_DEV FsAxis fs_axis(float coord, int n)
{
    const float x = coord * (float)n - 0.5f;
    float x0f = floorf(x);
    FsAxis r;
    r.f = x - x0f;
    r.w0 = 1.0f - r.f;
    if constexpr (GUARD)
        if (!(x0f > -1.0e9f && x0f < 1.0e9f)) x0f = 0.0f;
    r.i0 = (int)x0f;
    return r;
}

// p(x, row) * wx0 + p(x + 1, row) * fx from the two texels
PT_DEV float4 fs_row_mix(const float4 &p0, const float4 &p1, const FsAxis &ax)
{
    return make_float4(p0.x * ax.w0 + p1.x * ax.f, p0.y * ax.w0 + p1.y * ax.f, p0.z * ax.w0 + p1.z * ax.f,
                       p0.w * ax.w0 + p1.w * ax.f);
}
PT_DEV float4 fs_row_value_tex(const FsTile &t, const FsAxis &ax, int row)
{
    const size_t base = (size_t)wrapi(row, t.H) * t.W;
    return fs_row_mix(t.tex[base + wrapi(ax.i0, t.W)], t.tex[base + wrapi(ax.i0 + 1, t.W)], ax);
}
// the tile's texel pair at (ax.i0, row): index into t.lds
PT_DEV int fs_tile_index(const FsTile &t, const FsAxis &ax, int row)
{
    return (int)__umul24((unsigned)(row - t.y0), (unsigned)PT_FS_TW) + (ax.i0 - t.x0);
}
template <bool LDS>
PT_DEV float4 fs_row_value(const FsTile &t, const FsAxis &ax, int row)
{
    if constexpr (LDS) {
        const float4 *q = t.lds + fs_tile_index(t, ax, row);
        return fs_row_mix(q[0], q[1], ax);
    } else {
        return fs_row_value_tex(t, ax, row);
    }
}

PT_DEV float4 fs_combine(const float4 &ra, const float4 &rb, const FsAxis &ay)
{
    return make_float4(ra.x * ay.w0 + rb.x * ay.f, ra.y * ay.w0 + rb.y * ay.f, ra.z * ay.w0 + rb.z * ay.f,
                       ra.w * ay.w0 + rb.w * ay.f);
}

// Do all of this pixel's taps (and its centre sample) read texels of the tile?
PT_DEV bool denoise_inside(const DenoiseTaps *taps, const FsTile &t, float u, float v)
{
    struct Summary { float ox_lo, ox_hi, oy_lo, oy_hi; int finite; int pad[3]; };
    const Summary m = load_const_block(reinterpret_cast<const Summary *>(&taps->ox_lo));
    const float xa = fminf(fs_texel(u + m.ox_lo, t.W), fs_texel(u, t.W)), xb = fmaxf(fs_texel(u + m.ox_hi, t.W), fs_texel(u, t.W));
    const float ya = fminf(fs_texel(v + m.oy_lo, t.H), fs_texel(v, t.H)), yb = fmaxf(fs_texel(v + m.oy_hi, t.H), fs_texel(v, t.H));
    bool ok = m.finite != 0;
    ok = ok && xa >= (float)t.x0 && xb <= (float)(t.x0 + PT_FS_TW - 2);        // (a NaN fails the comparison)
    ok = ok && ya >= (float)t.y0 && yb <= (float)(t.y0 + PT_FS_TH - 2);
    return ok;
}

template <bool LDS>
PT_DEV float4 denoise(const DenoiseTaps *taps, const FsTile &t, float u, float v, float threshold)
{
    const float INV_SQRT_OF_2PI = 0.39894228040143267793994605993439f;
    const float inv_threshold_sqx2 = 0.5f / (threshold * threshold);
    const float inv_threshold_sqrt2pi = INV_SQRT_OF_2PI / threshold;
    float4 centr;
    {
        const FsAxis ax = fs_axis<!LDS>(u, t.W), ay = fs_axis<!LDS>(v, t.H);
        centr = fs_combine(fs_row_value<LDS>(t, ax, ay.i0), fs_row_value<LDS>(t, ax, ay.i0 + 1), ay);
    }
    float zbuff = 0.0f;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;      // (the sum's fourth component is not part of the pixel: fullscreen.wgsl:121)
#pragma unroll 1
    for (int c = 0; c < PT_DN_COLS; c++) {
        const DenoiseColumn C = load_const_block(&taps->col[c]);
        const FsAxis ax = fs_axis<!LDS>(u + C.ox, t.W);
        int prev = (int)0x80000000;               // row of the previous tap (none yet)
        float4 rb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int at = 0;                                // LDS: index of the lower row's texel pair
        auto tap = [&](float oy, float blur) {
            const FsAxis ay = fs_axis<!LDS>(v + oy, t.H);
            float4 ra = rb;
            if constexpr (LDS) {
                if (ay.i0 - 1 != prev) {
                    at = fs_tile_index(t, ax, ay.i0);
                    ra = fs_row_mix(t.lds[at], t.lds[at + 1], ax);
                }
                at += PT_FS_TW;
                rb = fs_row_mix(t.lds[at], t.lds[at + 1], ax);
            } else {
                if (ay.i0 - 1 != prev) ra = fs_row_value_tex(t, ax, ay.i0);
                rb = fs_row_value_tex(t, ax, ay.i0 + 1);
            }
            prev = ay.i0;
            const float4 walk = fs_combine(ra, rb, ay);
            const float dx = walk.x - centr.x, dy = walk.y - centr.y, dz = walk.z - centr.z, dw = walk.w - centr.w;
            const float dcdc = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
            const float delta = ptm::exp1_nonpos(-dcdc * inv_threshold_sqx2) * inv_threshold_sqrt2pi * blur;
            zbuff = zbuff + delta;
            sx = sx + delta * walk.x;
            sy = sy + delta * walk.y;
            sz = sz + delta * walk.z;
        };
        // (written out: the offsets and weights are scalar registers, which a loop cannot index)
#define PT_TAP(J) if (C.count > J) { tap(C.oy[J], C.blur[J]);
        PT_TAP(0) PT_TAP(1) PT_TAP(2) PT_TAP(3) PT_TAP(4) PT_TAP(5) PT_TAP(6) PT_TAP(7) PT_TAP(8) PT_TAP(9) PT_TAP(10) PT_TAP(11)
        }}}}}}}}}}}}
#undef PT_TAP
        static_assert(PT_DN_ROWS == 12, "taps written out above");
    }
    return make_float4(sx / zbuff, sy / zbuff, sz / zbuff, 1.0f);
}

// fullscreen.wgsl:88-103 (mat3x3f constructors are column-major)
PT_DEV f3 aces_tonemap(f3 c)
{
    const f3 v = F3((0.59719f * c.x + 0.35458f * c.y) + 0.04823f * c.z,
                    (0.07600f * c.x + 0.90834f * c.y) + 0.01566f * c.z,
                    (0.02840f * c.x + 0.13383f * c.y) + 0.83777f * c.z);
    f3 r;
    {
        const float a = v.x * (v.x + 0.0245786f) - 0.000090537f;
        const float b = v.x * (0.983729f * v.x + 0.4329510f) + 0.238081f;
        r.x = a / b;
    }
    {
        const float a = v.y * (v.y + 0.0245786f) - 0.000090537f;
        const float b = v.y * (0.983729f * v.y + 0.4329510f) + 0.238081f;
        r.y = a / b;
    }
    {
        const float a = v.z * (v.z + 0.0245786f) - 0.000090537f;
        const float b = v.z * (0.983729f * v.z + 0.4329510f) + 0.238081f;
        r.z = a / b;
    }
    const float mx = (1.60475f * r.x + -0.53108f * r.y) + -0.07367f * r.z;
    const float my = (-0.10208f * r.x + 1.10813f * r.y) + -0.00605f * r.z;
    const float mz = (-0.00327f * r.x + -0.07276f * r.y) + 1.07602f * r.z;
    // vec3f(1.0 / 2.2): an AbstractFloat const-expression, evaluated in double, then rounded (fullscreen.wgsl:102)
    const float g = (float)(1.0 / 2.2);
    return F3(ptm::pow1(clamp1(mx, 0.0f, 1.0f), g), ptm::pow1(clamp1(my, 0.0f, 1.0f), g),
              ptm::pow1(clamp1(mz, 0.0f, 1.0f), g));
}

PT_DEV uint32_t to_unorm8(float v)
{
    float q = clamp1(v, 0.0f, 1.0f);
    if (q != q) q = 0.0f;
    return (uint32_t)rintf(q * 255.0f);
}

// fragmentMain, fullscreen.wgsl:109-132, one thread per canvas pixel; canvas row 0
// is the top (framebuffer order), quad uv (0,0) sits at clip (-1,-1) = bottom left.
__global__ void __launch_bounds__(256) k_fullscreen(const FsUniforms fs, const float4 *__restrict__ tex,
                                                    int tex_w, int tex_h, int canvas_w, int canvas_h,
                                                    const DenoiseTaps *__restrict__ taps,
                                                    float4 *__restrict__ out_f32, uint32_t *__restrict__ out_rgba8)
{
    __shared__ float4 tile[PT_FS_TW * PT_FS_TH];
    const int px = blockIdx.x * 16 + (threadIdx.x & 15);
    const int py = blockIdx.y * 16 + (threadIdx.x >> 4);
    const float u = (((float)px + 0.5f) / (float)canvas_w) * fs.scaling;
    const float v = (1.0f - ((float)py + 0.5f) / (float)canvas_h) * fs.scaling;
    float4 c4;
    if (fs.denoise == 1u) {
        // The tile: from PT_FS_MARGIN texels left of / below the texel pair of the block's first column / last row (v runs against
        // py).  A guess that is right for every scaling <= 1; where it is not, the wave reads the texture.
        FsTile t;
        t.lds = tile; t.tex = tex; t.W = tex_w; t.H = tex_h;
        t.x0 = fs_axis<true>((((float)(blockIdx.x * 16) + 0.5f) / (float)canvas_w) * fs.scaling, tex_w).i0 - PT_FS_MARGIN;
        t.y0 = fs_axis<true>((1.0f - ((float)(blockIdx.y * 16 + 15) + 0.5f) / (float)canvas_h) * fs.scaling, tex_h).i0 - PT_FS_MARGIN;
        for (int i = (int)threadIdx.x; i < PT_FS_TW * PT_FS_TH; i += 256) {
            const int ty = i / PT_FS_TW, tx = i - ty * PT_FS_TW;
            tile[i] = tex[(size_t)wrapi(t.y0 + ty, tex_h) * tex_w + wrapi(t.x0 + tx, tex_w)];
        }
        __syncthreads();
        if (px >= canvas_w || py >= canvas_h) return;
        if (__builtin_amdgcn_ballot_w64(!denoise_inside(taps, t, u, v)) == 0) c4 = denoise<true>(taps, t, u, v, 0.08f);
        else c4 = denoise<false>(taps, t, u, v, 0.08f);
    } else {
        if (px >= canvas_w || py >= canvas_h) return;
        c4 = sample_repeat(tex, tex_w, tex_h, u, v);
    }
    f3 c = F3(c4.x, c4.y, c4.z);
    if (fs.tonemapping == 1u) c = aces_tonemap(c);
    else if (fs.tonemapping == 2u) c = F3(c.x / (c.x + 1.0f), c.y / (c.y + 1.0f), c.z / (c.z + 1.0f));
    const size_t i = (size_t)py * canvas_w + px;
    if (out_f32) out_f32[i] = make_float4(c.x, c.y, c.z, 1.0f);
    if (out_rgba8) out_rgba8[i] = to_unorm8(c.x) | (to_unorm8(c.y) << 8) | (to_unorm8(c.z) << 16) | 0xff000000u;
}

// `taps`: fullscreen_taps_bytes() of device memory owned by the caller; `taps_current`: the table in it was built by an
// earlier call on this stream from the same res_x / res_y (the setup kernel is skipped).
void launch_fullscreen(const FsUniforms &fs, const float4 *tex, int tex_w, int tex_h, int canvas_w,
                       int canvas_h, void *taps, bool taps_current, float4 *out_f32, uint32_t *out_rgba8, hipStream_t s)
{
    if (canvas_w <= 0 || canvas_h <= 0) return;
    if (fs.denoise == 1u && !taps_current)
        hipLaunchKernelGGL(k_fullscreen_setup, dim3(1), dim3(64), 0, s, fs, static_cast<DenoiseTaps *>(taps));
    const dim3 grid((canvas_w + 15) / 16, (canvas_h + 15) / 16), block(256);
    hipLaunchKernelGGL(k_fullscreen, grid, block, 0, s, fs, tex, tex_w, tex_h, canvas_w, canvas_h,
                       static_cast<const DenoiseTaps *>(taps), out_f32, out_rgba8);
}

// ---------------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------------
template <int VARIANT>
__global__ void __launch_bounds__(64) k_debug_intersect(const SceneRefs sc, const float *__restrict__ rays,
                                                        size_t n, float *__restrict__ out)
{
    __shared__ uint32_t stack_lds[PT_MAX_STACK * 64];
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const f3 o = F3(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
    const f3 d = F3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    Counters cnt = { 0, 0, 0, 0, 0, 0, 0, 0 };
    Best best;
    traverse<VARIANT>(sc, o, d, stack_lds + threadIdx.x, best, cnt);
    float *r = out + i * 12;
    f3 position = F3(0.0f, 0.0f, 0.0f), normal = F3(0.0f, 0.0f, 0.0f);
    int32_t mi = -1;
    if (best.tri >= 0) finish_hit(sc, o, d, best, position, normal, mi);
    r[0] = best.tri >= 0 ? 1.0f : 0.0f;
    r[1] = best.t;
    r[2] = position.x; r[3] = position.y; r[4] = position.z;
    r[5] = normal.x; r[6] = normal.y; r[7] = normal.z;
    r[8] = (float)mi;
    r[9] = (float)cnt.box; r[10] = (float)cnt.tri; r[11] = (float)cnt.overflow;
}

void launch_debug_intersect(const SceneRefs &scene, const float *rays, size_t n, float *out, int variant,
                            hipStream_t s)
{
    if (n == 0) return;
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    if (variant >= 3) hipLaunchKernelGGL(k_debug_intersect<3>, grid, block, 0, s, scene, rays, n, out);
    else if (variant == 2) hipLaunchKernelGGL(k_debug_intersect<2>, grid, block, 0, s, scene, rays, n, out);
    else hipLaunchKernelGGL(k_debug_intersect<1>, grid, block, 0, s, scene, rays, n, out);
}

__global__ void __launch_bounds__(256) k_debug_math(int fn, const float *__restrict__ a,
                                                    const float *__restrict__ b, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b ? b[i] : 0.0f;
    float r;
    switch (fn) {
    case 0: r = ptm::sin1(x); break;
    case 1: r = ptm::cos1(x); break;
    case 2: r = ptm::tan1(x); break;
    case 3: r = ptm::log1(x); break;
    case 4: r = ptm::exp1(x); break;
    case 5: r = ptm::atan2_1(x, y); break;
    case 6: r = ptm::asin1(x); break;
    case 7: r = ptm::pow1(x, y); break;
    case 8: r = ptm::round_f16(x); break;
    case 9: r = sqrtf(x); break;
    case 11: r = ptm::sqrt_exact(x); break;
    case 12: r = ptm::rcp_exact(x); break;
    case 13: r = normalize(F3(x, y, x - y)).x; break;
    case 14: r = normalize(F3(x, y, x - y)).y; break;
    case 15: r = normalize(F3(x, y, x - y)).z; break;
    case 16: r = ptm::log1_unit(x); break;           // (only for x = 0 or x in [2^-32, 1]: what rand() returns)
    default: r = x / y; break;
    }
    out[i] = r;
}

// What the launch that has just run COST per ray, for the host's choice of the next launch's build (pt_context.hip: adapt_walk): the
// counter set's box tests and rays, summed over its blocks, as the difference to the previous call for that set.  A reset of the
// counters in between (sums below the previous ones) restarts the difference.  One block; microseconds.
__global__ void __launch_bounds__(256) k_walk_stats(const uint64_t *__restrict__ counters, int nblocks, uint64_t *prev, uint64_t *out, uint64_t seq)
{
    __shared__ uint64_t sb[256], sr[256];
    uint64_t b = 0, r = 0;
    for (int i = (int)threadIdx.x; i < nblocks; i += 256) {
        b += counters[(size_t)i * CNT_COUNT + CNT_BOX];
        r += counters[(size_t)i * CNT_COUNT + CNT_RAYS];
    }
    sb[threadIdx.x] = b; sr[threadIdx.x] = r;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sb[threadIdx.x] += sb[threadIdx.x + s]; sr[threadIdx.x] += sr[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint64_t tb = sb[0], tr = sr[0], pb = prev[0], pr = prev[1];
        const bool restarted = tb < pb || tr < pr;
        prev[0] = tb; prev[1] = tr;
        __hip_atomic_store(out + 1, restarted ? tb : tb - pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(out + 2, restarted ? tr : tr - pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(out + 0, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void launch_walk_stats(const uint64_t *counters, int nblocks, uint64_t *prev, uint64_t *out, uint64_t seq, hipStream_t s)
{
    if (nblocks <= 0 || !counters || !prev || !out) return;
    hipLaunchKernelGGL(k_walk_stats, dim3(1), dim3(256), 0, s, counters, nblocks, prev, out, seq);
}

__global__ void __launch_bounds__(256) k_patch_cull(float4 *__restrict__ packets, const uint32_t *__restrict__ cull, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) packets[(size_t)i * 4 + 3].w = __uint_as_float(cull[i]);
}

void launch_patch_cull(float4 *packets, const uint32_t *cull, uint32_t npackets, hipStream_t s)
{
    if (npackets == 0) return;
    hipLaunchKernelGGL(k_patch_cull, dim3((npackets + 255u) / 256u), dim3(256), 0, s, packets, cull, npackets);
}

// the three position vectors (48 of 112 bytes) of every triangle record, packed: what the context's cull analysis reads back
__global__ void __launch_bounds__(256) k_pack_vertices(const float4 *__restrict__ tris, float4 *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < 3; k++) out[(size_t)i * 3 + k] = tris[(size_t)i * 7 + k];
}

void launch_pack_vertices(const float4 *tris, float4 *out, uint32_t ntris, hipStream_t s)
{
    if (ntris == 0) return;
    hipLaunchKernelGGL(k_pack_vertices, dim3((ntris + 255u) / 256u), dim3(256), 0, s, tris, out, ntris);
}

void launch_debug_math(int fn, const float *a, const float *b, float *out, size_t n, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, a, b, out, n);
}

}  // namespace pt
