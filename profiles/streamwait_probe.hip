// Probe: can a stream wait (hipStreamWaitValue32) on a word that a RUNNING kernel of another
// stream writes?  Used to decide how a queued raytrace launch is held back until the previous
// launch's work queue has run empty.   hipcc --offload-arch=gfx950 -O2 streamwait_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_busy_then_signal(uint32_t *flag, uint64_t *stamp, unsigned long long spin_ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(32);
    stamp[0] = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t1 < spin_ticks) __builtin_amdgcn_s_sleep(32);     // keep running after the signal
    stamp[1] = __builtin_amdgcn_s_memrealtime();
}

__global__ void k_stamp(uint64_t *stamp) { stamp[2] = __builtin_amdgcn_s_memrealtime(); }

int main()
{
    uint32_t *flag = nullptr;
    uint64_t *stamp = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
    std::printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 2;
    CHECK(hipMalloc((void **)&stamp, 64));
    CHECK(hipMemset(flag, 0, 8));
    CHECK(hipMemset(stamp, 0, 64));
    hipStream_t a, b;
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_busy_then_signal, dim3(1), dim3(64), 0, a, flag, stamp, 200000ull);     // 2 ms + 2 ms
    e = hipStreamWaitValue32(b, flag, 1u, hipStreamWaitValueGte, 0xffffffffu);
    std::printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { (void)hipDeviceSynchronize(); return 3; }
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, b, stamp);
    CHECK(hipStreamSynchronize(b));
    CHECK(hipStreamSynchronize(a));
    uint64_t h[3];
    CHECK(hipMemcpy(h, stamp, sizeof h, hipMemcpyDeviceToHost));
    std::printf("signal at t=0, waiter ran at t=%+.1f us, signalling kernel ended at t=%+.1f us -> %s\n",
                ((double)h[2] - (double)h[0]) / 100.0, ((double)h[1] - (double)h[0]) / 100.0,
                (h[2] >= h[0] && h[2] < h[1]) ? "the wait was released by the running kernel" : "NOT released mid-kernel");
    return 0;
}
