// log_unit_proof.hip -- ptm::log1_unit against ptm::log1 on the device, for EVERY value rand() can return
// (raytrace.wgsl:253-259: f32(r) / f32(4294967295.0) = RN(r) * 2^-32 for each of the 2^32 integers r), and, beyond that,
// for every binary32 in the function's stated domain: +0 and [2^-32, 1].
// The claim (comment above log1_unit in csrc/pt_devmath.h): the two return the same bits.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//        -o /tmp/log_unit_proof profiles/log_unit_proof.hip
// run:   /tmp/log_unit_proof                                    (log: profiles/r04_w_log_unit_proof.log)
#include "../webgpu-pathtracer_amd/csrc/pt_devmath.h"
#include <cstdio>

__global__ void k_proof(unsigned long long *out)
{
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long tested = 0, wrong = 0, infs = 0, dom = 0, dom_wrong = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += T) {
        // (a) the integer r of rand(): x = f32(r) / 2^32
        const uint32_t r = (uint32_t)b;
        const float x = (float)r / 4294967296.0f;
        const uint32_t want = __float_as_uint(ptm::log1(x)), got = __float_as_uint(ptm::log1_unit(x));
        tested++;
        if (want != got) wrong++;
        if (want == 0xff800000u) infs++;
        // (b) the bit pattern b as a float, where it lies in the domain
        const float y = __uint_as_float((uint32_t)b);
        if ((uint32_t)b == 0u || (y >= 2.3283064365386963e-10f && y <= 1.0f)) {
            dom++;
            if (__float_as_uint(ptm::log1(y)) != __float_as_uint(ptm::log1_unit(y))) dom_wrong++;
        }
    }
    atomicAdd(&out[0], tested); atomicAdd(&out[1], wrong); atomicAdd(&out[2], infs); atomicAdd(&out[3], dom); atomicAdd(&out[4], dom_wrong);
}

int main()
{
    unsigned long long *d = nullptr, h[5] = { 0, 0, 0, 0, 0 };
    if (hipMalloc((void **)&d, sizeof h) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(k_proof, dim3(4096), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("log1_unit vs log1: %llu values of rand()'s integer, %llu disagreements (%llu of them give -inf); "
           "%llu floats of the domain {+0} u [2^-32, 1], %llu disagreements\n", h[0], h[1], h[2], h[3], h[4]);
    return (h[1] == 0 && h[4] == 0) ? 0 : 1;
}
