// probe_block_placement.hip -- where does the dispatcher put N one-wave workgroups that stay resident (a persistent
// launch narrower than the machine)?  Each workgroup records its XCC / SE / CU (s_getreg HW_ID, XCC_ID) and then waits
// ~300 us, so that all N are resident together; 7 680 B of LDS and at most five waves per SIMD, like the raytrace
// kernel's tuned instantiation.  Prints waves per CU: min / mean / max over the CUs used, CUs used, and per XCC.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_block_placement profiles/probe_block_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) k_place(uint32_t *out, long long wait_cycles)
{
    __shared__ uint32_t lds[7680 / 4];
    lds[threadIdx.x] = threadIdx.x;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);        // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);       // HW_REG_XCC_ID
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < wait_cycles) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc + lds[0]; }
}

int main()
{
    uint32_t *d = nullptr;
    if (hipMalloc((void **)&d, 8192 * 8) != hipSuccess) { printf("no device\n"); return 2; }
    for (int n : { 5120, 4096, 3072, 2560, 2048, 1024 }) {
        hipLaunchKernelGGL(k_place, dim3(n), dim3(64), 0, 0, d, 30000LL);       // wall clock: 100 MHz -> 300 us
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
        std::vector<uint32_t> h(n * 2);
        (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        std::map<uint32_t, int> per_cu, per_xcc;
        for (int i = 0; i < n; i++) {
            const uint32_t hw = h[i * 2], xcc = h[i * 2 + 1] & 15u;
            const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
            per_xcc[xcc]++;
        }
        int mn = 1 << 30, mx = 0;
        for (auto &kv : per_cu) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
        printf("%5d workgroups: %3zu CUs used, waves per CU min %d mean %.1f max %d; per XCC:", n, per_cu.size(), mn, (double)n / per_cu.size(), mx);
        for (auto &kv : per_xcc) printf(" %d", kv.second);
        printf("\n");
    }
    return 0;
}
