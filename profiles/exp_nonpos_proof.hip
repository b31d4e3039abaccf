// exp_nonpos_proof.hip -- ptm::exp1_nonpos against ptm::exp1 on the device, for EVERY input the de-noise pass can hand
// it: all 2^31 floats with the sign bit set (-0 .. -inf and the negative-sign NaNs), +0, and all positive-sign NaNs.
// The claim (comment above exp1_nonpos in csrc/pt_devmath.h): the two return the same bits.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//        -o /tmp/exp_nonpos_proof profiles/exp_nonpos_proof.hip
// run:   /tmp/exp_nonpos_proof                                  (log: profiles/r03_j_exp_nonpos_proof.log)
#include "../webgpu-pathtracer_amd/csrc/pt_devmath.h"
#include <cstdio>

__global__ void k_proof(unsigned long long *out)
{
    // thread t checks bit patterns t, t + T, ... of the 2^32; patterns of positive non-NaN numbers other than +0 are
    // outside the function's domain and skipped
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long tested = 0, wrong = 0, zeros = 0, subnormal = 0, nans = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += T) {
        const uint32_t u = (uint32_t)b;
        const float x = __uint_as_float(u);
        const bool is_nan = x != x;
        if (!(is_nan || x <= 0.0f)) continue;
        const uint32_t want = __float_as_uint(ptm::exp1(x)), got = __float_as_uint(ptm::exp1_nonpos(x));
        tested++;
        if (want != got) wrong++;
        if (is_nan) nans++;
        else if (want == 0u) zeros++;
        else if ((want >> 23) == 0u) subnormal++;
    }
    atomicAdd(&out[0], tested); atomicAdd(&out[1], wrong); atomicAdd(&out[2], zeros); atomicAdd(&out[3], subnormal);
    atomicAdd(&out[4], nans);
}

int main()
{
    unsigned long long *d = nullptr, h[5] = { 0, 0, 0, 0, 0 };
    if (hipMalloc((void **)&d, sizeof h) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(k_proof, dim3(4096), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("exp1_nonpos vs exp1: %llu inputs (every x <= 0, +0, every NaN), %llu disagreements; results: %llu zeros, %llu subnormals, %llu NaNs\n",
           h[0], h[1], h[2], h[3], h[4]);
    return h[1] == 0 ? 0 : 1;
}
