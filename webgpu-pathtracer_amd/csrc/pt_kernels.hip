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
//   k_accumulate               accumulate.wgsl computeMain
//   k_fullscreen               fullscreen.wgsl fragmentMain (de-noise + tone-map)
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
// pinned: normalize(v) = v / sqrt(dot(v, v))
PT_DEV f3 normalize(f3 a)
{
    const float l = sqrtf(dot(a, a));
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
    const float f = 1.0f / det;
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
            if (ray_triangle(o, d, xyz(pa), xyz(pb), xyz(pc), t, u, v) && t < best.t) {
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

template <int VARIANT>
PT_DEV void traverse(const SceneRefs &sc, const f3 &o, const f3 &d, uint32_t *stack, Best &best,
                     Counters &cnt)
{
    if (VARIANT == 2) traverse_packets(sc, o, d, stack, best, cnt);
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
    const float rho = sqrtf(-2.0f * ptm::log1(rand1(seed)));
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
    const float rho = sqrtf(rand1(seed));
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

PT_DEV int local_to_global_row(int ly, const Tile &t)
{
    const int b = ly / t.block_rows;
    return (b * t.nranks + t.rank) * t.block_rows + (ly - b * t.block_rows);
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
                const f3 env = sample_env(sc.env, sc.env_w, sc.env_h, u, v);
                light = light + (ray_color * env) * un.env_intensity;
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

    Counters cnt = { 0, 0, 0, 0, 0, 0, 0 };
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

int raytrace_grid_blocks(const Tile &tile)
{
    const int tiles_x = (tile.tex_w + 7) / 8;
    const int tiles_y = (tile.local_rows + 7) / 8;
    return tiles_x * tiles_y;
}

void launch_raytrace(const RtLaunch &L, bool fuse, int variant, hipStream_t s)
{
    const int blocks = raytrace_grid_blocks(L.tile);
    if (blocks <= 0) return;
    const dim3 grid(blocks), block(64);
    if (variant == 2) {
        if (fuse) hipLaunchKernelGGL((k_raytrace<true, 2>), grid, block, 0, s, L);
        else hipLaunchKernelGGL((k_raytrace<false, 2>), grid, block, 0, s, L);
    } else {
        if (fuse) hipLaunchKernelGGL((k_raytrace<true, 1>), grid, block, 0, s, L);
        else hipLaunchKernelGGL((k_raytrace<false, 1>), grid, block, 0, s, L);
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
PT_DEV int wrapi(int v, int n) { int m = v % n; return m < 0 ? m + n : m; }

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

// fullscreen.wgsl:53-86 with sigma 5, kSigma 1, threshold 0.08 (:117-119)
PT_DEV float4 denoise(const float4 *tex, int W, int H, float res_x, float res_y, float u, float v,
                      float sigma, float k_sigma, float threshold)
{
    const float INV_PI = 0.31830988618379067153776752674503f;
    const float INV_SQRT_OF_2PI = 0.39894228040143267793994605993439f;
    const float radius = rintf(k_sigma * sigma);
    const float rad_q = radius * radius;
    const float inv_sigma_qx2 = 0.5f / (sigma * sigma);
    const float inv_sigma_qx2pi = INV_PI * inv_sigma_qx2;
    const float inv_threshold_sqx2 = 0.5f / (threshold * threshold);
    const float inv_threshold_sqrt2pi = INV_SQRT_OF_2PI / threshold;
    const float4 centr = sample_repeat(tex, W, H, u, v);
    float zbuff = 0.0f;
    float4 abuff = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (float x = -radius; x <= radius; x = x + 1.0f) {
        const float pt = sqrtf(rad_q - x * x);
        for (float y = -pt; y <= pt; y = y + 1.0f) {
            const float dd = x * x + y * y;
            const float blur = ptm::exp1(-dd * inv_sigma_qx2) * inv_sigma_qx2pi;
            const float4 walk = sample_repeat(tex, W, H, u + x / res_x, v + y / res_y);
            const float dx = walk.x - centr.x, dy = walk.y - centr.y, dz = walk.z - centr.z, dw = walk.w - centr.w;
            const float dcdc = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
            const float delta = ptm::exp1(-dcdc * inv_threshold_sqx2) * inv_threshold_sqrt2pi * blur;
            zbuff = zbuff + delta;
            abuff.x = abuff.x + delta * walk.x;
            abuff.y = abuff.y + delta * walk.y;
            abuff.z = abuff.z + delta * walk.z;
            abuff.w = abuff.w + delta * walk.w;
        }
    }
    return make_float4(abuff.x / zbuff, abuff.y / zbuff, abuff.z / zbuff, abuff.w / zbuff);
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
    const float g = 1.0f / 2.2f;
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
                                                    float4 *__restrict__ out_f32, uint32_t *__restrict__ out_rgba8)
{
    const int px = blockIdx.x * 16 + (threadIdx.x & 15);
    const int py = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (px >= canvas_w || py >= canvas_h) return;
    const float u = (((float)px + 0.5f) / (float)canvas_w) * fs.scaling;
    const float v = (1.0f - ((float)py + 0.5f) / (float)canvas_h) * fs.scaling;
    float4 c4 = sample_repeat(tex, tex_w, tex_h, u, v);
    if (fs.denoise == 1u) c4 = denoise(tex, tex_w, tex_h, fs.res_x, fs.res_y, u, v, 5.0f, 1.0f, 0.08f);
    f3 c = F3(c4.x, c4.y, c4.z);
    if (fs.tonemapping == 1u) c = aces_tonemap(c);
    else if (fs.tonemapping == 2u) c = F3(c.x / (c.x + 1.0f), c.y / (c.y + 1.0f), c.z / (c.z + 1.0f));
    const size_t i = (size_t)py * canvas_w + px;
    if (out_f32) out_f32[i] = make_float4(c.x, c.y, c.z, 1.0f);
    if (out_rgba8) out_rgba8[i] = to_unorm8(c.x) | (to_unorm8(c.y) << 8) | (to_unorm8(c.z) << 16) | 0xff000000u;
}

void launch_fullscreen(const FsUniforms &fs, const float4 *tex, int tex_w, int tex_h, int canvas_w,
                       int canvas_h, float4 *out_f32, uint32_t *out_rgba8, hipStream_t s)
{
    if (canvas_w <= 0 || canvas_h <= 0) return;
    const dim3 grid((canvas_w + 15) / 16, (canvas_h + 15) / 16), block(256);
    hipLaunchKernelGGL(k_fullscreen, grid, block, 0, s, fs, tex, tex_w, tex_h, canvas_w, canvas_h, out_f32,
                       out_rgba8);
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
    Counters cnt = { 0, 0, 0, 0, 0, 0, 0 };
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
    if (variant == 2) hipLaunchKernelGGL(k_debug_intersect<2>, grid, block, 0, s, scene, rays, n, out);
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
    default: r = x / y; break;
    }
    out[i] = r;
}

void launch_debug_math(int fn, const float *a, const float *b, float *out, size_t n, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, a, b, out, n);
}

}  // namespace pt
