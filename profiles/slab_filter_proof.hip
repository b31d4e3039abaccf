// slab_filter_proof.hip -- the filtered slab test of the WIDE walk against the exact test, on the device, with the
// kernel's OWN functions (the kernel file is included as it stands): slab_q0 / slab_margin / slab_hit versus
// ray_aabb_fast (which tests/test_gpu_parity.py holds bit-identical to the reference's plain-division test).
//
// Claim checked (proof: comment above slab_q0 in csrc/pt_kernels.hip): for every ray and box the fast path admits,
//     slab_margin(tmin~, tmax~) > 0   ==>   slab_hit(tmin~, tmax~) == ray_aabb_fast(...)
// Pairs are drawn from a counter-based generator and are mostly ADVERSARIAL: the ray is aimed (in double precision,
// then rounded) at a corner, an edge point or a face point of the box, so that tmin and tmax agree to rounding; a
// third of the origins sit on a face of the box (tmax or tmin == 0 up to rounding); a sixth of the boxes are flat on
// one or two axes; directions are un-normalised by powers of two and sign-flipped; scales run over 2^-20 .. 2^20.
// Every 64th pair is an ordinary random one (the rate of undecided boxes there is what the filter costs).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//        -o /tmp/slab_filter_proof profiles/slab_filter_proof.hip
// run:   /tmp/slab_filter_proof [pairs, default 1e11]          (log: profiles/r03_a_slab_filter_proof.log)
#include "../webgpu-pathtracer_amd/csrc/pt_kernels.hip"
#include <cstdio>
#include <cstdlib>

using namespace pt;

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t s;
    __device__ uint64_t next() { s = mix64(s); return s; }
    __device__ double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }       // [0, 1)
    __device__ double sym() { return uni() * 2.0 - 1.0; }
};

__global__ void k_proof(uint64_t base, uint64_t per_thread, unsigned long long *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long tested = 0, decided = 0, wrong = 0, hits = 0, ordinary = 0, ordinary_undecided = 0;
    for (uint64_t it = 0; it < per_thread; it++) {
        Rng r;
        r.s = mix64(base + tid * per_thread + it);
        const uint64_t sel = r.next();
        const double scale = exp2((double)((int)(sel & 63) % 41 - 20));
        double c[3], h[3];
        for (int k = 0; k < 3; k++) {
            c[k] = r.sym() * scale;
            h[k] = fabs(r.sym()) * scale * exp2(-(double)((sel >> (8 + 4 * k)) & 15));
        }
        float mn[3], mx[3];
        for (int k = 0; k < 3; k++) { mn[k] = (float)(c[k] - h[k]); mx[k] = (float)(c[k] + h[k]); }
        const unsigned flat = (unsigned)((sel >> 24) & 31);       // 1..3: that axis flat; 4..5: two axes flat
        if (flat >= 1 && flat <= 3) mx[flat - 1] = mn[flat - 1];
        if (flat == 4) { mx[0] = mn[0]; mx[1] = mn[1]; }
        if (flat == 5) { mx[1] = mn[1]; mx[2] = mn[2]; }
        const bool plain = ((sel >> 32) & 63) == 0;
        double tgt[3], org[3];
        for (int k = 0; k < 3; k++) {
            const unsigned pick = (unsigned)((sel >> (40 + 2 * k)) & 3);
            tgt[k] = pick == 0 ? (double)mn[k] : (pick == 1 ? (double)mx[k] : (double)mn[k] + r.uni() * ((double)mx[k] - (double)mn[k]));
            org[k] = tgt[k] + r.sym() * scale * 4.0;
        }
        if (((sel >> 48) & 3) == 0) {          // origin on a face of the box
            for (int k = 0; k < 3; k++) org[k] = (double)mn[k] + r.uni() * ((double)mx[k] - (double)mn[k]);
            const int ax = (int)((sel >> 50) & 3) % 3;
            org[ax] = ((sel >> 52) & 1) ? (double)mn[ax] : (double)mx[ax];
        }
        if (plain) for (int k = 0; k < 3; k++) { tgt[k] = r.sym() * scale * 3.0; org[k] = r.sym() * scale * 3.0; }
        const f3 o = F3((float)org[0], (float)org[1], (float)org[2]);
        double dd[3], nrm = 0.0;
        for (int k = 0; k < 3; k++) { dd[k] = tgt[k] - (double)(k == 0 ? o.x : (k == 1 ? o.y : o.z)); nrm += dd[k] * dd[k]; }
        nrm = sqrt(nrm);
        if (!(nrm > 0.0)) continue;
        const double len = exp2((double)((int)((sel >> 54) & 7) - 2)) * (((sel >> 57) & 1) ? -1.0 : 1.0);
        const f3 d = F3((float)(dd[0] / nrm * len), (float)(dd[1] / nrm * len), (float)(dd[2] / nrm * len));
        // the fast path's admission (kernel: RayPre::flags bit 3, the packet's box flags)
        const RayPre pre = ray_prepare(o, d, 1u);
        if (pre.flags & 8u) continue;
        bool box_ok = true;
        for (int k = 0; k < 3; k++) box_ok = box_ok && safe_magnitude(mn[k]) && safe_magnitude(mx[k]);
        if (!box_ok) continue;
        const bool exact = ray_aabb_fast(o, d, pre, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]);
        f3 tn;
        float tf;
        slab_q0(o, pre, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], tn, tf);
        const float key = fmaxf(fmaxf(tn.x, tn.y), tn.z);
        const bool dec = slab_margin(key, tf) > 0.0f;
        const bool approx = slab_hit(key, tf);
        tested++;
        hits += exact ? 1 : 0;
        decided += dec ? 1 : 0;
        wrong += (dec && approx != exact) ? 1 : 0;
        if (plain) { ordinary++; ordinary_undecided += dec ? 0 : 1; }
    }
    atomicAdd(out + 0, tested); atomicAdd(out + 1, decided); atomicAdd(out + 2, wrong); atomicAdd(out + 3, hits);
    atomicAdd(out + 4, ordinary); atomicAdd(out + 5, ordinary_undecided);
}

int main(int argc, char **argv)
{
    const double want = argc > 1 ? atof(argv[1]) : 1e11;
    const int blocks = 256 * 32, threads = 256;
    const uint64_t per_thread = 4096;
    const uint64_t per_launch = (uint64_t)blocks * threads * per_thread;
    unsigned long long *d_out = nullptr, h[6] = { 0, 0, 0, 0, 0, 0 };
    if (hipMalloc((void **)&d_out, sizeof h) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    hipMemset(d_out, 0, sizeof h);
    uint64_t base = 0x1234567ull;
    double done = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    while (done < want) {
        hipLaunchKernelGGL(k_proof, dim3(blocks), dim3(threads), 0, 0, base, per_thread, d_out);
        base += per_launch;
        done += (double)per_launch;
    }
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 3; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    printf("pairs generated %.3e; admitted to the fast path %llu; exact-test hits %llu\n", done, h[0], h[3]);
    printf("decided by the filter %llu (%.4f %%), undecided %llu\n", h[1], 100.0 * (double)h[1] / (double)h[0], h[0] - h[1]);
    printf("ordinary pairs %llu, undecided among them %llu (%.3g)\n", h[4], h[5], h[4] ? (double)h[5] / (double)h[4] : 0.0);
    printf("DECIDED BOXES THAT DISAGREE WITH THE EXACT TEST: %llu\n", h[2]);
    printf("%.1f s\n", ms * 1e-3);
    return h[2] == 0 ? 0 : 1;
}
