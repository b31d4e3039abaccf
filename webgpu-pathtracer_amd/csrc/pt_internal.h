// pt_internal.h -- small helpers shared by the translation units of libmi3pt.so
#pragma once
#include <string>

// Records `msg` as the calling thread's last error and returns `code`.
int pt_set_error(int code, const std::string &msg);

// pt_lbvh.hip: device-side linear BVH over n 112-byte triangle records at d_tris; writes (2n - 1)
// 48-byte node records to the HOST buffer nodes_out.  Returns 0, or -1 with `err` set.
#include <hip/hip_runtime.h>
namespace pt {
int lbvh_build(const void *d_tris, size_t n, void *nodes_out, float *build_ms, hipStream_t stream, std::string &err);
}

// ---- host-side helpers shared by pt_context.hip and pt_host_wide.cpp (plain C++: the CPU sanitizer builds cover them)
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>
#include "pt_kernels.h"
#include "../../include/mi3pt.h"
static inline float ldf(const uint8_t *p, size_t off) { float f; std::memcpy(&f, p + off, 4); return f; }
static inline int32_t ldi(const uint8_t *p, size_t off) { int32_t v; std::memcpy(&v, p + off, 4); return v; }
static inline uint32_t ldu(const uint8_t *p, size_t off) { uint32_t v; std::memcpy(&v, p + off, 4); return v; }
// precondition of the exact fast slab test (pt_kernels.hip, RayPre): per box, every coordinate is 0 or within [2^-70, 2^60]
bool node_box_safe(const uint8_t *src, size_t node);
// the SAH-optimal grouping of a binary tree's nodes into W-wide packets, and the eight-wide packets of kernel variant 14 (pt_host_wide.cpp)
struct WideCollapse {
    int W = 4;
    std::vector<uint8_t> k0;         // per node: how many of its packet's W entries go to the left child's side
    std::vector<uint8_t> split;      // [node][i], i = 2 .. W-1: entries for the left side when the node is opened into i entries; 0 = "as with i - 1"
    const uint8_t *src = nullptr;
    bool leaf(size_t i) const { return ldi(src + i * MI3PT_BVHNODE_STRIDE, 28) == 1; }
    int32_t left(size_t i) const { return ldi(src + i * MI3PT_BVHNODE_STRIDE, 32); }
    int32_t right(size_t i) const { return ldi(src + i * MI3PT_BVHNODE_STRIDE, 36); }
    // the entries of node x's packet (binary node indices)
    void children_of(size_t x, std::vector<int32_t> &out) const
    {
        out.clear();
        std::vector<std::pair<int32_t, int>> work;      // (node, entries it may use)
        work.emplace_back(right(x), W - (int)k0[x]);
        work.emplace_back(left(x), (int)k0[x]);
        while (!work.empty()) {
            auto [c, i] = work.back();
            work.pop_back();
            if (leaf((size_t)c) || i <= 1) { out.push_back(c); continue; }
            int k = 0;
            while (i >= 2 && (k = split[(size_t)c * (size_t)W + (size_t)i]) == 0) i--;
            if (i < 2) { out.push_back(c); continue; }
            work.emplace_back(right((size_t)c), i - k);
            work.emplace_back(left((size_t)c), k);
        }
    }
};
bool collapse_optimal(const uint8_t *src, size_t n, int W, const std::vector<uint8_t> *closed, WideCollapse &out);
struct Cw8Build {
    std::vector<pt::CW8Packet> packets;
    std::vector<pt::TriPacket64> records;
    int height = 0;             // levels of packets: the walk's node stack never holds more entries (one per level)
    double mean_children = 0.0;
};
bool build_cw8(const uint8_t *src, size_t n, const float *verts /* 12 floats per triangle: a, pad, b, pad, c, pad */, size_t nt,
               const std::vector<float> &wmax, Cw8Build &out, bool greedy = false);
