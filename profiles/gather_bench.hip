// Micro-benchmark: how fast can a CU gather one 64-byte record per lane?
//
// The raytrace kernel's node step makes every lane fetch its own 64-byte node packet with four
// global_load_dwordx4 (256 lane-addresses per wave-step, four to the same cache line).  This
// program times that access pattern against alternatives with the same grid shape as the
// kernel (one-wave workgroups, 16 per CU, persistent):
//   own4   each lane loads its own packet: 4 x dwordx4                      (what the kernel does)
//   own1   each lane loads only the first 16 bytes of its packet: 1 x dwordx4 (address-rate probe)
//   quad   the 4 lanes of a quad load one packet's four 16-byte chunks together (one coalesced
//          64-byte segment per quad and instruction), 4 instructions cover the quad's 4 packets,
//          chunks are exchanged through LDS (conflict-free swizzle) and read back by the owner
// for a table that fits L2 (2048 packets = 128 KiB) and one that does not (4 Mi packets = 256 MiB).
//
//   hipcc --offload-arch=gfx950 -O3 profiles/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t &s)
{
    s = s * 747796405u + 2891336453u;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (r >> 22) ^ r;
}

__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

template <int MODE>
__global__ void __launch_bounds__(64, 4) k_gather(const float4 *__restrict__ table, uint32_t mask, int iters,
                                                  float *__restrict__ out, unsigned long long *cycles)
{
    __shared__ float4 xch[64 * 4];          // 4 KiB exchange buffer (MODE 2)
    const int lane = threadIdx.x;
    uint32_t seed = blockIdx.x * 64u + lane + 12345u;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const uint32_t ref = pcg(seed) & mask;
        if (MODE == 0) {
            const float4 *p = table + (size_t)ref * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += (sum4(a) + sum4(b)) + (sum4(c) + sum4(d));
        } else if (MODE == 1) {
            const float4 a = table[(size_t)ref * 4];
            acc += sum4(a);
        } else if (MODE == 2) {
            // quad-cooperative: instruction i fetches the packet of quad lane i; this lane takes chunk k
            const int k = lane & 3, q4 = lane & ~3;
            float4 got[4];
            // quad_perm broadcast of quad lane i (DPP: a plain VALU move, no LDS traffic)
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)ref, 0x00, 0xf, 0xf, true);
            const uint32_t r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)ref, 0x55, 0xf, 0xf, true);
            const uint32_t r2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)ref, 0xaa, 0xf, 0xf, true);
            const uint32_t r3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)ref, 0xff, 0xf, 0xf, true);
            got[0] = table[(size_t)r0 * 4 + k];
            got[1] = table[(size_t)r1 * 4 + k];
            got[2] = table[(size_t)r2 * 4 + k];
            got[3] = table[(size_t)r3 * 4 + k];
            // exchange: chunk c of packet p lives at row c, slot 4*((p+c)&3) + ((p>>2)&3) of its 16-packet group
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int p = q4 + i;
                xch[(p >> 4) * 64 + k * 16 + 4 * ((p + k) & 3) + ((p >> 2) & 3)] = got[i];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): one-wave workgroup, no barrier needed
            float4 mine[4];
#pragma unroll
            for (int c = 0; c < 4; c++)
                mine[c] = xch[(lane >> 4) * 64 + c * 16 + 4 * ((lane + c) & 3) + ((lane >> 2) & 3)];
            acc += (sum4(mine[0]) + sum4(mine[1])) + (sum4(mine[2]) + sum4(mine[3]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static int run(const char *name, const float4 *table, uint32_t mask, int iters, float *out, unsigned long long *cyc,
               int blocks, double *checksum)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(64), 0, 0, table, mask, iters / 8, out, cyc);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(64), 0, 0, table, mask, iters, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    std::vector<float> ho((size_t)blocks * 64);
    CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    double mean = 0.0, cs = 0.0;
    for (auto c : h) mean += (double)c;
    for (auto v : ho) cs += v;
    mean /= blocks;
    const double bytes = (double)blocks * 64 * iters * (MODE == 1 ? 16.0 : 64.0);
    std::printf("%-6s table %8u packets: %8.3f ms  %7.1f shader cycles per wave-step  %7.1f GB/s  checksum %.6e\n", name,
                mask + 1, ms, mean / iters, bytes / ms * 1e-6, cs);
    *checksum = cs;
    return 0;
}

int main()
{
    const int blocks = 256 * 16, iters = 4000;
    const uint32_t big = 1u << 22;
    float4 *table;
    float *out;
    unsigned long long *cyc;
    CHECK(hipMalloc((void **)&table, (size_t)big * 64));
    CHECK(hipMalloc((void **)&out, (size_t)blocks * 64 * 4));
    CHECK(hipMalloc((void **)&cyc, blocks * sizeof(unsigned long long)));
    std::vector<float> h((size_t)big * 16);
    uint32_t s = 1u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f); }
    CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (uint32_t n : { 2048u, big }) {
        double c0, c1, c2;
        if (run<0>("own4", table, n - 1, iters, out, cyc, blocks, &c0)) return 1;
        if (run<1>("own1", table, n - 1, iters, out, cyc, blocks, &c1)) return 1;
        if (run<2>("quad", table, n - 1, iters, out, cyc, blocks, &c2)) return 1;
        std::printf("  own4 == quad: %s\n", c0 == c2 ? "yes" : "NO");
    }
    return 0;
}
