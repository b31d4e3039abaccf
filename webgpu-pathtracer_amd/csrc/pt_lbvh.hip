// pt_lbvh.hip -- a linear BVH built on the device ("GPU/LBVH-style parallel builder
// as an alternative tree for the 10 M-triangle case").
//
// NOT the reference's tree: RaytracePass.buildBVH (raytrace.ts:540-655) is a top-down full-sweep SAH
// build, reproduced bit for bit by mi3pt_host_build_bvh_f64.  This builder trades tree quality for
// build time (milliseconds instead of seconds on millions of triangles) and writes the SAME 48-byte
// BVHNode records (raytrace.wgsl:51-64), so everything downstream -- mi3pt_upload_bvh, the packets,
// every kernel, the oracle -- takes its output unchanged.  The reference walk has no culling, so the
// closest hit (and hence the image) does not depend on the tree except where two triangles are hit
// at exactly the same t.
//
// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (HPG 2012):
// 63-bit Morton codes of the triangle centroids (21 bits per axis; equal codes are told apart by their
// position after the sort), radix-sorted with the triangle index as payload (hipCUB); one thread per
// internal node finds its range and split from the common prefixes; boxes are fitted bottom-up; nodes are numbered in pre-order (index = 2 * first leaf of
// the range + number of left turns on the path from the root), so that a child always follows its
// parent -- the order mi3pt_upload_bvh validates.
#include "pt_internal.h"
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdint>
#include <string>

namespace pt {

__device__ __forceinline__ unsigned long long expand21(unsigned long long v)      // 21 bits -> every third bit
{
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__device__ __forceinline__ int order_f(float f)                // monotone float -> int map for atomicMin/Max
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float unorder_f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

struct Box { float mn[3], mx[3]; };

// per triangle: box and centroid; scene bounds of the centroids
__global__ void __launch_bounds__(256) k_lbvh_prims(const float4 *__restrict__ tris, uint32_t n, Box *__restrict__ tri_box,
                                                    float *__restrict__ centroid, int *__restrict__ bounds /* 6 ordered ints */)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 a = tris[(size_t)i * 7 + 0], b = tris[(size_t)i * 7 + 1], c = tris[(size_t)i * 7 + 2];
    Box bx;
    bx.mn[0] = fminf(fminf(a.x, b.x), c.x); bx.mx[0] = fmaxf(fmaxf(a.x, b.x), c.x);
    bx.mn[1] = fminf(fminf(a.y, b.y), c.y); bx.mx[1] = fmaxf(fmaxf(a.y, b.y), c.y);
    bx.mn[2] = fminf(fminf(a.z, b.z), c.z); bx.mx[2] = fmaxf(fmaxf(a.z, b.z), c.z);
    tri_box[i] = bx;
    for (int k = 0; k < 3; k++) {
        const float ce = 0.5f * bx.mn[k] + 0.5f * bx.mx[k];
        centroid[(size_t)i * 3 + k] = ce;
        if (ce == ce) {
            atomicMin(bounds + k, order_f(ce));
            atomicMax(bounds + 3 + k, order_f(ce));
        }
    }
}

__global__ void __launch_bounds__(256) k_lbvh_keys(const float *__restrict__ centroid, uint32_t n, const int *__restrict__ bounds,
                                                   unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = unorder_f(bounds[k]), hi = unorder_f(bounds[3 + k]);
        const float extent = hi - lo;
        float t = extent > 0.0f ? (centroid[(size_t)i * 3 + k] - lo) / extent : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        if (!(t == t)) t = 0.0f;
        q[k] = (unsigned long long)fminf(t * 2097152.0f, 2097151.0f);
    }
    keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);       // 63-bit Morton code
    vals[i] = i;
}

// length of the common prefix of the (code, position) pairs: equal codes are told apart by position
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    return x != 0ull ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// node ids while building: internal i in [0, n-2], leaf k as (n - 1 + k)
__global__ void __launch_bounds__(256) k_lbvh_hierarchy(const unsigned long long *__restrict__ keys, int n, int *__restrict__ left,
                                                        int *__restrict__ right, int *__restrict__ first, int *__restrict__ last,
                                                        int *__restrict__ parent, uint8_t *__restrict__ is_left)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int lc = (lo == gamma) ? (n - 1 + gamma) : gamma;
    const int rc = (hi == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
    left[i] = lc; right[i] = rc; first[i] = lo; last[i] = hi;
    parent[lc] = i; is_left[lc] = 1;
    parent[rc] = i; is_left[rc] = 0;
    if (i == 0) { parent[0] = -1; is_left[0] = 0; }
}

// bottom-up fit: the second child to arrive at a node unions the two boxes and climbs on
__global__ void __launch_bounds__(256) k_lbvh_fit(const uint32_t *__restrict__ tri_of, int n, const Box *__restrict__ tri_box,
                                                  const int *__restrict__ left, const int *__restrict__ right,
                                                  const int *__restrict__ parent, Box *__restrict__ node_box, int *__restrict__ arrived)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    int node = n - 1 + k;
    node_box[node] = tri_box[tri_of[k]];
    __threadfence();
    int p = parent[node];
    while (p >= 0) {
        if (atomicAdd(arrived + p, 1) == 0) return;               // the sibling will finish this node
        __threadfence();
        const Box a = node_box[left[p]], b = node_box[right[p]];
        Box u;
        for (int c = 0; c < 3; c++) { u.mn[c] = fminf(a.mn[c], b.mn[c]); u.mx[c] = fmaxf(a.mx[c], b.mx[c]); }
        node_box[p] = u;
        __threadfence();
        p = parent[p];
    }
}

// pre-order position of every node and the 48-byte record the reference's buffer holds there
__global__ void __launch_bounds__(256) k_lbvh_emit(const uint32_t *__restrict__ tri_of, int n, const int *__restrict__ left,
                                                   const int *__restrict__ first, const int *__restrict__ last,
                                                   const int *__restrict__ parent, const uint8_t *__restrict__ is_left,
                                                   const Box *__restrict__ node_box, uint8_t *__restrict__ out)
{
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= 2 * n - 1) return;
    const bool leaf = node >= n - 1;
    const int a = leaf ? node - (n - 1) : first[node];
    int turns = 0;
    for (int x = node; parent[x] >= 0; x = parent[x]) turns += is_left[x];
    const int idx = 2 * a + turns;
    float *f = reinterpret_cast<float *>(out + (size_t)idx * 48);
    int *w = reinterpret_cast<int *>(out + (size_t)idx * 48);
    const Box b = node_box[node];
    f[0] = b.mn[0]; f[1] = b.mn[1]; f[2] = b.mn[2]; w[3] = 0;
    f[4] = b.mx[0]; f[5] = b.mx[1]; f[6] = b.mx[2];
    if (leaf) {
        w[7] = 1; w[8] = -1; w[9] = -1; w[10] = (int)tri_of[a];
    } else {
        const int lc = left[node];
        const int left_leaves = lc >= n - 1 ? 1 : last[lc] - first[lc] + 1;
        w[7] = 0; w[8] = idx + 1; w[9] = idx + 2 * left_leaves; w[10] = -1;
    }
    w[11] = 0;
}

#define LB_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); goto done; } } while (0)

// tris: device pointer to n 112-byte records; nodes_out: HOST buffer of (2n-1) * 48 bytes.
int lbvh_build(const void *d_tris, size_t n, void *nodes_out, float *build_ms, hipStream_t stream, std::string &err)
{
    if (n == 0) { err = "no triangles"; return -1; }
    const int N = (int)n;
    Box *tri_box = nullptr, *node_box = nullptr;
    float *centroid = nullptr;
    int *bounds = nullptr, *left = nullptr, *right = nullptr, *first = nullptr, *last = nullptr, *parent = nullptr, *arrived = nullptr;
    uint8_t *is_left = nullptr, *d_out = nullptr;
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    uint32_t *vals = nullptr, *vals_sorted = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t nodes = 2 * n - 1;
    const unsigned blocks_n = (unsigned)((n + 255) / 256), blocks_nodes = (unsigned)((nodes + 255) / 256);
    const int init_bounds[6] = { 0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000 };
    float ms = 0.0f;
    LB_TRY(hipMalloc((void **)&tri_box, n * sizeof(Box)));
    LB_TRY(hipMalloc((void **)&node_box, nodes * sizeof(Box)));
    LB_TRY(hipMalloc((void **)&centroid, n * 12));
    LB_TRY(hipMalloc((void **)&bounds, 24));
    LB_TRY(hipMalloc((void **)&keys, n * 8));
    LB_TRY(hipMalloc((void **)&keys_sorted, n * 8));
    LB_TRY(hipMalloc((void **)&vals, n * 4));
    LB_TRY(hipMalloc((void **)&vals_sorted, n * 4));
    LB_TRY(hipMalloc((void **)&left, n * 4));
    LB_TRY(hipMalloc((void **)&right, n * 4));
    LB_TRY(hipMalloc((void **)&first, n * 4));
    LB_TRY(hipMalloc((void **)&last, n * 4));
    LB_TRY(hipMalloc((void **)&parent, nodes * 4));
    LB_TRY(hipMalloc((void **)&arrived, n * 4));
    LB_TRY(hipMalloc((void **)&is_left, nodes));
    LB_TRY(hipMalloc((void **)&d_out, nodes * 48));
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_sorted, vals, vals_sorted, N, 0, 63, stream));
    LB_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    LB_TRY(hipEventCreate(&e0));
    LB_TRY(hipEventCreate(&e1));
    LB_TRY(hipMemcpyAsync(bounds, init_bounds, 24, hipMemcpyHostToDevice, stream));
    LB_TRY(hipMemsetAsync(arrived, 0, n * 4, stream));
    LB_TRY(hipMemsetAsync(parent, 0xff, nodes * 4, stream));
    LB_TRY(hipMemsetAsync(is_left, 0, nodes, stream));
    LB_TRY(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(k_lbvh_prims, dim3(blocks_n), dim3(256), 0, stream, static_cast<const float4 *>(d_tris), (uint32_t)n, tri_box, centroid, bounds);
    hipLaunchKernelGGL(k_lbvh_keys, dim3(blocks_n), dim3(256), 0, stream, centroid, (uint32_t)n, bounds, keys, vals);
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_sorted, vals, vals_sorted, N, 0, 63, stream));
    if (n > 1)
        hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(blocks_n), dim3(256), 0, stream, keys_sorted, N, left, right, first, last, parent, is_left);
    hipLaunchKernelGGL(k_lbvh_fit, dim3(blocks_n), dim3(256), 0, stream, vals_sorted, N, tri_box, left, right, parent, node_box, arrived);
    hipLaunchKernelGGL(k_lbvh_emit, dim3(blocks_nodes), dim3(256), 0, stream, vals_sorted, N, left, first, last, parent, is_left, node_box, d_out);
    LB_TRY(hipGetLastError());
    LB_TRY(hipEventRecord(e1, stream));
    LB_TRY(hipMemcpyAsync(nodes_out, d_out, nodes * 48, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    LB_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (build_ms) *build_ms = ms;
done:
    for (void *p : { (void *)tri_box, (void *)node_box, (void *)centroid, (void *)bounds, (void *)keys, (void *)keys_sorted, (void *)vals, (void *)vals_sorted, (void *)left,
                     (void *)right, (void *)first, (void *)last, (void *)parent, (void *)arrived, (void *)is_left, (void *)d_out, tmp })
        if (p) (void)hipFree(p);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return err.empty() ? 0 : -1;
}

}  // namespace pt
