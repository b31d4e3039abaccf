// Calibration of rocprofv3's FETCH_SIZE for THIS kernel's access patterns (round-4 verdict, weak #3; guide
// MI355X_MICROARCH.md:297-301: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// k_raytrace_sm's reads are divergent gathers: every lane fetches its own 64-byte node packet / 64-byte triangle record
// (four global_load_dwordx4), 48-byte triangle packets in the exact-packet variants (three), 16-byte environment texels.
// Each kernel below reads a table with one of those shapes, EVERY RECORD EXACTLY ONCE (a bijection of the record index, so the
// byte count is known and no cache can serve a record twice), from a table far larger than the Infinity Cache (4 GiB by default);
// run under `rocprofv3 --pmc FETCH_SIZE` (and the raw TCC_EA0_RDREQ counters) the ratio known bytes / FETCH_SIZE is the factor
// bench.py has to apply.  `stream16` is the control the guide's x2 was measured on.  The same gathers over a 128 MiB table
// (every record read 32 times: cache-resident after the first sweep, like the 181 MB dragon scene) show whether Infinity-Cache
// hits are counted.  The program also prints the useful GB/s each pattern reaches: the HBM roof OF THAT ACCESS PATTERN.
//
//   hipcc --offload-arch=gfx950 -O3 profiles/fetch_calibration.hip -o /tmp/fetch_calibration
//   /tmp/fetch_calibration [table MiB = 4096] [small table MiB = 128]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// a bijection of [0, 2^bits): odd multiplies and xor-shifts are each invertible modulo 2^bits
__device__ __forceinline__ uint32_t scramble(uint32_t i, uint32_t bits)
{
    const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    i = (i * 0x9E3779B1u) & mask;
    i ^= i >> (bits / 2 + 1);
    i = (i * 0x85EBCA6Bu) & mask;
    i ^= i >> (bits / 2 - 1);
    i = (i * 0xC2B2AE35u) & mask;
    return i;
}

__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

// CHUNKS x 16 bytes per record, records STRIDE16 x 16 bytes apart; record r of 2^bits, each read once
template <int CHUNKS, int STRIDE16>
__global__ void __launch_bounds__(64) k_gather(const float4 *__restrict__ table, uint32_t bits, uint32_t sweeps, float *__restrict__ out)
{
    const uint32_t threads = gridDim.x * 64u;
    const uint32_t tid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t records = 1u << bits;
    float acc = 0.0f;
    for (uint32_t s = 0; s < sweeps; s++)
        for (uint32_t i = tid; i < records; i += threads) {
            const float4 *p = table + (size_t)scramble(i ^ (s * 0x5bd1e995u & (records - 1u)), bits) * STRIDE16;
            float4 v[CHUNKS];
#pragma unroll
            for (int c = 0; c < CHUNKS; c++) v[c] = p[c];
#pragma unroll
            for (int c = 0; c < CHUNKS; c++) acc += sum4(v[c]);
        }
    if (acc == 123.456f) out[tid] = acc;          // never true for the table's contents: keeps the loads alive
}

// the control: wide coalesced streaming read, 16 B per lane
__global__ void __launch_bounds__(64) k_stream16(const float4 *__restrict__ table, size_t n16, float *__restrict__ out)
{
    const size_t threads = (size_t)gridDim.x * 64u;
    const size_t tid = (size_t)blockIdx.x * 64u + threadIdx.x;
    float acc = 0.0f;
    for (size_t i = tid; i < n16; i += threads) acc += sum4(table[i]);
    if (acc == 123.456f) out[tid] = acc;
}

static uint32_t log2_floor(size_t v) { uint32_t b = 0; while ((v >> (b + 1)) != 0) b++; return b; }

template <typename F>
static int timed(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    std::printf("%-28s requested %12.0f bytes  %9.3f ms  %8.1f GB/s useful\n", name, bytes, ms, bytes / ms / 1e6);
    std::fflush(stdout);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t big_mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096;
    const size_t small_mib = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 128;
    const size_t big = big_mib << 20, small = small_mib << 20;
    float4 *table = nullptr;
    float *out = nullptr;
    CHECK(hipMalloc(&table, big));
    CHECK(hipMemset(table, 0, big));
    const int grid = 256 * 32;          // 32 one-wave workgroups per CU
    CHECK(hipMalloc(&out, (size_t)grid * 64 * sizeof(float)));
    CHECK(hipDeviceSynchronize());
    std::printf("table %zu MiB (small %zu MiB), grid %d one-wave workgroups\n", big_mib, small_mib, grid);

    // large table: every record exactly once
    const uint32_t b64 = log2_floor(big / 64), b128 = log2_floor(big / 128), b48 = log2_floor(big / 48), b16 = log2_floor(big / 16);
    if (timed("stream16_big", (double)big, [&] { k_stream16<<<grid, 64>>>(table, big / 16, out); })) return 1;
    if (timed("gather64_big", 64.0 * (1ull << b64), [&] { k_gather<4, 4><<<grid, 64>>>(table, b64, 1, out); })) return 1;
    if (timed("gather48_big", 48.0 * (1ull << b48), [&] { k_gather<3, 3><<<grid, 64>>>(table, b48, 1, out); })) return 1;
    if (timed("gather128_big", 128.0 * (1ull << b128), [&] { k_gather<8, 8><<<grid, 64>>>(table, b128, 1, out); })) return 1;
    // 16-byte texels: a quarter of them (distinct ones, scattered over the whole table)
    if (timed("gather16_of64_big", 16.0 * (1ull << b64), [&] { k_gather<1, 4><<<grid, 64>>>(table, b64, 1, out); })) return 1;
    if (timed("gather16_all_big", 16.0 * (1ull << b16), [&] { k_gather<1, 1><<<grid, 64>>>(table, b16, 1, out); })) return 1;
    // first 32 bytes of every 64-byte record (what a leaf-box-only look-up would read)
    if (timed("gather32_of64_big", 32.0 * (1ull << b64), [&] { k_gather<2, 4><<<grid, 64>>>(table, b64, 1, out); })) return 1;

    // small table: 32 sweeps, each record once per sweep -- resident in the Infinity Cache (and partly in L2) after the first
    const uint32_t s64 = log2_floor(small / 64), s48 = log2_floor(small / 48);
    const uint32_t sweeps = 32;
    if (timed("stream16_small_x32", (double)small * sweeps, [&] { for (uint32_t s = 0; s < sweeps; s++) k_stream16<<<grid, 64>>>(table, small / 16, out); })) return 1;
    if (timed("gather64_small_x32", 64.0 * (1ull << s64) * sweeps, [&] { k_gather<4, 4><<<grid, 64>>>(table, s64, sweeps, out); })) return 1;
    if (timed("gather48_small_x32", 48.0 * (1ull << s48) * sweeps, [&] { k_gather<3, 3><<<grid, 64>>>(table, s48, sweeps, out); })) return 1;
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(table));
    CHECK(hipFree(out));
    return 0;
}
