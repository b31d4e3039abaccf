// Exhaustive check of the division-free slab quotients used by the raytrace kernel.
//
// For binary32 n, d (normal, and no intermediate under/overflow -- the kernel's guards) the
// sequence
//      y  = RN(1/d)                 (IEEE division, once per ray and axis)
//      q0 = RN(n*y)
//      q1 = RN(q0 + RN(n - d*q0)*y)         3 VALU ops per quotient   ("short" form)
//      q2 = RN(q1 + RN(n - d*q1)*y)         5 VALU ops per quotient   ("long" form, Markstein)
// is compared with RN(n/d) for EVERY pair of significands (2^23 x 2^23).  All operations are
// round-to-nearest and commute with scaling by powers of two and with sign changes, so the
// significand pairs in [1,2) x [1,2) cover every input the guards admit.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         profiles/div_proof.hip -o /tmp/div_proof && /tmp/div_proof [first_d_chunk] [n_chunks]
//
// Output: mismatch counts of the short and long forms (and of q0 alone, to show the check has
// teeth), and the d significands for which the short form fails, if any.
//
// `div_proof rcp` instead checks, for ALL 2^32 binary32 inputs d, whether one Newton step on the
// hardware reciprocal,  y0 = v_rcp_f32(d);  y1 = RN(y0 + RN(1 - d*y0)*y0),  equals RN(1/d)
// (the IEEE division the compiler expands to 11 instructions), and prints the magnitude range
// of d over which it always does.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Result {
    unsigned long long bad_short, bad_long, bad_q0;
    unsigned int nbad_d;
    unsigned int bad_d[256];
    unsigned int bad_n[256];
};

// one thread per d significand of the chunk; loops over all 2^23 n significands
__global__ void __launch_bounds__(256) k_check(uint32_t d_first, Result *res)
{
    const uint32_t dm = d_first + blockIdx.x * blockDim.x + threadIdx.x;
    const float d = __uint_as_float(0x3f800000u | dm);
    const float y = 1.0f / d;
    unsigned int bs = 0, bl = 0, b0 = 0, first_n = 0xffffffffu;
#pragma unroll 4
    for (uint32_t nm = 0; nm < (1u << 23); nm++) {
        const float n = __uint_as_float(0x3f800000u | nm);
        const float want = n / d;
        const float q0 = n * y;
        const float q1 = fmaf(fmaf(-d, q0, n), y, q0);
        const float q2 = fmaf(fmaf(-d, q1, n), y, q1);
        b0 += (q0 != want);
        const bool s = (q1 != want);
        if (s && first_n == 0xffffffffu) first_n = nm;
        bs += s;
        bl += (q2 != want);
    }
    if (bs | bl | b0) {
        atomicAdd(&res->bad_short, (unsigned long long)bs);
        atomicAdd(&res->bad_long, (unsigned long long)bl);
        atomicAdd(&res->bad_q0, (unsigned long long)b0);
    }
    if (bs) {
        const unsigned int slot = atomicAdd(&res->nbad_d, 1u);
        if (slot < 256) { res->bad_d[slot] = dm; res->bad_n[slot] = first_n; }
    }
}

struct RcpResult {
    unsigned long long bad, bad_raw;
    unsigned int min_bad_abs, max_bad_abs;     // magnitude bits of failing |d| inside [2^-64, 2^64]
    unsigned long long bad_inside;
    unsigned int first_bad[16];
    unsigned int nfirst;
};

__global__ void __launch_bounds__(256) k_check_rcp(uint32_t first, RcpResult *res)
{
    const uint32_t bits = first + blockIdx.x * blockDim.x + threadIdx.x;
    const float d = __uint_as_float(bits);
    const float want = 1.0f / d;
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float y1 = fmaf(fmaf(-d, y0, 1.0f), y0, y0);
    const bool nan_both = (want != want) && (y1 != y1);
    if (y0 != want && !((want != want) && (y0 != y0))) atomicAdd(&res->bad_raw, 1ull);
    if (y1 != want && !nan_both) {
        atomicAdd(&res->bad, 1ull);
        const uint32_t mag = bits & 0x7fffffffu;
        if (mag >= 0x1f800000u && mag <= 0x5f800000u) {        // 2^-64 .. 2^64
            atomicAdd(&res->bad_inside, 1ull);
            atomicMin(&res->min_bad_abs, mag);
            atomicMax(&res->max_bad_abs, mag);
            const unsigned int slot = atomicAdd(&res->nfirst, 1u);
            if (slot < 16) res->first_bad[slot] = bits;
        }
    }
}

struct SqrtResult { unsigned long long bad_raw, bad_a, bad_b, bad_a_in, bad_b_in; unsigned int first_a[8], first_b[8], na, nb; };

// all positive floats: candidates for a cheaper correctly rounded sqrt
//   A: s0 = v_sqrt_f32(x); s1 = fma(fma(-s0, s0, x) * 0.5, v_rcp_f32(s0), s0)
//   B: y = v_rsq_f32(x); s0 = x*y; s1 = fma(fma(-s0, s0, x), 0.5*y, s0)
__global__ void __launch_bounds__(256) k_check_sqrt(uint32_t first, SqrtResult *res)
{
    const uint32_t bits = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (bits >= 0x7f800000u) return;
    const float x = __uint_as_float(bits);
    const float want = sqrtf(x);
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float a = fmaf(fmaf(-s0, s0, x) * 0.5f, __builtin_amdgcn_rcpf(s0), s0);
    const float y = __builtin_amdgcn_rsqf(x);
    const float t0 = x * y;
    const float b = fmaf(fmaf(-t0, t0, x), 0.5f * y, t0);
    const bool inside = bits >= 0x1f800000u && bits <= 0x5f800000u;       // 2^-64 .. 2^64
    if (s0 != want) atomicAdd(&res->bad_raw, 1ull);
    if (a != want) {
        atomicAdd(&res->bad_a, 1ull);
        if (inside) { atomicAdd(&res->bad_a_in, 1ull); const unsigned int k = atomicAdd(&res->na, 1u); if (k < 8) res->first_a[k] = bits; }
    }
    if (b != want) {
        atomicAdd(&res->bad_b, 1ull);
        if (inside) { atomicAdd(&res->bad_b_in, 1ull); const unsigned int k = atomicAdd(&res->nb, 1u); if (k < 8) res->first_b[k] = bits; }
    }
}

static int check_sqrt()
{
    SqrtResult *dres, h{};
    CHECK(hipMalloc((void **)&dres, sizeof(SqrtResult)));
    CHECK(hipMemset(dres, 0, sizeof(SqrtResult)));
    for (uint32_t c = 0; c < 128; c++) {
        hipLaunchKernelGGL(k_check_sqrt, dim3((1u << 24) / 256), dim3(256), 0, 0, c << 24, dres);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(&h, dres, sizeof(SqrtResult), hipMemcpyDeviceToHost));
    std::printf("RESULT sqrt: all non-negative finite inputs; v_sqrt_f32 alone differs from RN(sqrt x) for %llu; "
                "form A (sqrt, residual, rcp) %llu differ, %llu with x in [2^-64, 2^64]; form B (rsq) %llu differ, %llu in range\n",
                h.bad_raw, h.bad_a, h.bad_a_in, h.bad_b, h.bad_b_in);
    for (unsigned int i = 0; i < h.na && i < 8; i++) std::printf("  form A fails for x bits 0x%08x\n", h.first_a[i]);
    for (unsigned int i = 0; i < h.nb && i < 8; i++) std::printf("  form B fails for x bits 0x%08x\n", h.first_b[i]);
    return 0;
}

static int check_rcp()
{
    RcpResult *dres, h{};
    h.min_bad_abs = 0xffffffffu;
    CHECK(hipMalloc((void **)&dres, sizeof(RcpResult)));
    CHECK(hipMemcpy(dres, &h, sizeof(RcpResult), hipMemcpyHostToDevice));
    for (uint32_t c = 0; c < 256; c++) {
        hipLaunchKernelGGL(k_check_rcp, dim3((1u << 24) / 256), dim3(256), 0, 0, c << 24, dres);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(&h, dres, sizeof(RcpResult), hipMemcpyDeviceToHost));
    std::printf("RESULT rcp: all 2^32 inputs; v_rcp_f32 alone differs from RN(1/d) for %llu; after one Newton step %llu differ, "
                "%llu of them with |d| in [2^-64, 2^64]\n", h.bad_raw, h.bad, h.bad_inside);
    for (unsigned int i = 0; i < h.nfirst && i < 16; i++) std::printf("  failing d bits 0x%08x\n", h.first_bad[i]);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "rcp") return check_rcp();
    if (argc > 1 && std::string(argv[1]) == "sqrt") return check_sqrt();
    const uint32_t chunk = 1u << 18;                       // d significands per launch
    const uint32_t nchunks_all = (1u << 23) / chunk;       // 32
    uint32_t first = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 0;
    uint32_t count = argc > 2 ? (uint32_t)std::atoi(argv[2]) : nchunks_all - first;
    if (first >= nchunks_all) return 2;
    if (first + count > nchunks_all) count = nchunks_all - first;
    Result *dres, h{};
    CHECK(hipMalloc((void **)&dres, sizeof(Result)));
    CHECK(hipMemset(dres, 0, sizeof(Result)));
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t c = first; c < first + count; c++) {
        hipLaunchKernelGGL(k_check, dim3(chunk / 256), dim3(256), 0, 0, c * chunk, dres);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&h, dres, sizeof(Result), hipMemcpyDeviceToHost));
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("d chunk %u/%u done (%.1f s): pairs %.4g  mismatches q0 %llu  short(3 ops) %llu  long(5 ops) %llu  failing d %u\n",
                    c + 1, nchunks_all, s, (double)(c - first + 1) * chunk * (double)(1u << 23), h.bad_q0, h.bad_short,
                    h.bad_long, h.nbad_d);
        std::fflush(stdout);
    }
    for (unsigned int i = 0; i < h.nbad_d && i < 256; i++)
        std::printf("short form fails for d significand 0x%06x (first n significand 0x%06x)\n", h.bad_d[i], h.bad_n[i]);
    std::printf("RESULT pairs %.6g  q0 %llu  short %llu  long %llu  failing_d %u\n",
                (double)count * chunk * (double)(1u << 23), h.bad_q0, h.bad_short, h.bad_long, h.nbad_d);
    return 0;
}
