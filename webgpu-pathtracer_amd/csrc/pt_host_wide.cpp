// pt_host_wide.cpp -- host-side construction of the walks' wide packets from the reference's binary tree (plain C++, no device calls):
//   collapse_optimal               which descendants of a node its W-wide packet holds (SAH-optimal dynamic programme)
//   build_cw8                      the eight-wide compressed packets + triangle records of kernel variant 14
//   mi3pt_host_eight_wide_check    builds them and verifies the result independently of the builder (what the CPU tests call)
// Called by the scene analysis (pt_context.hip: prepare_cull); covered by the CPU sanitizer builds (tests/tools/sanitize_cpu.sh).
#include "../../include/mi3pt.h"
#include "pt_internal.h"

#include <algorithm>
#include <cmath>
#include <string>
#include <utility>

bool node_box_safe(const uint8_t *src, size_t node)
{
    const uint8_t *r = src + node * MI3PT_BVHNODE_STRIDE;
    for (size_t off : { (size_t)0, (size_t)4, (size_t)8, (size_t)16, (size_t)20, (size_t)24 }) {
        const float v = ldf(r, off);
        const float a = v < 0 ? -v : v;
        if (!(v == 0.0f || (a >= 8.470329472543003e-22f && a <= 1.152921504606847e18f))) return false;
    }
    return true;
}

// ---- Which descendants of a binary node become the children of its W-wide packet: the SAH-optimal collapse (Ylitie, Karras, Laine 2017,
// section 3.1, with one triangle per leaf).  cost(n, i) = the least expected number of packet visits for the subtree under n when it is
// represented by at most i packets-or-leaves hanging under ONE parent packet (surface area relative to the root = the chance that a
// random ray visits):
//     cost(leaf, i) = 0;   cost(n, 1) = area(n) + min_{0<k<W} cost(left, k) + cost(right, W - k)           (n becomes a packet)
//     cost(n, i)    = min( cost(n, i - 1),  min_{0<k<i} cost(left, k) + cost(right, i - k) )               (n is opened: its children stand in for it)
// bottom-up over the nodes (children have larger indices than their parent: checked at upload).  A node that must not be opened (a box that
// does not contain its children's: `closed`) only has cost(n, 1).  The greedy collapse of rounds 2 - 5 (open the child with the largest area
// until the packet is full) fills the packets near the root and leaves the bottom of the tree in packets of two: 3.0 children per 4-ary packet
// and 4.0 per 8-ary one on the 870 k-triangle scene; the optimal one packs what can be packed.
bool collapse_optimal(const uint8_t *src, size_t n, int W, const std::vector<uint8_t> *closed, WideCollapse &out)
{
    out.W = W; out.src = src;
    out.k0.assign(n, 1);
    out.split.assign(n * (size_t)W, 0);
    std::vector<float> cost(n * (size_t)W, 0.0f);        // [node][i], i = 1 .. W-1 (slot 0 unused)
    auto area = [&](size_t i) {
        const uint8_t *r = src + i * MI3PT_BVHNODE_STRIDE;
        const double x = (double)ldf(r, 16) - ldf(r, 0), y = (double)ldf(r, 20) - ldf(r, 4), z = (double)ldf(r, 24) - ldf(r, 8);
        const double a = x * y + x * z + y * z;
        return a == a && a > 0.0 ? a : 0.0;
    };
    const double a0 = area(0) > 0.0 ? area(0) : 1.0;
    for (size_t i = n; i-- > 0;) {
        if (out.leaf(i)) continue;                        // cost 0 for every i
        const int32_t l = out.left(i), r = out.right(i);
        if (l < 0 || r < 0 || (size_t)l >= n || (size_t)r >= n || (size_t)l <= i || (size_t)r <= i) return false;
        const float *cl = &cost[(size_t)l * (size_t)W], *cr = &cost[(size_t)r * (size_t)W];
        float *c = &cost[i * (size_t)W];
        // as a packet of its own
        float best = 3.0e38f;
        int bk = 1;
        for (int k = 1; k < W; k++) {
            const float v = cl[k] + cr[W - k];
            if (v < best) { best = v; bk = k; }
        }
        out.k0[i] = (uint8_t)bk;
        c[1] = (float)(area(i) / a0) + best;
        const bool may_open = !(closed && (*closed)[i]);
        for (int j = 2; j < W; j++) {
            float v = c[j - 1];
            int vk = 0;
            if (may_open)
                for (int k = 1; k < j; k++) {
                    const float d = cl[k] + cr[j - k];
                    if (d < v) { v = d; vk = k; }
                }
            c[j] = v;
            out.split[i * (size_t)W + (size_t)j] = (uint8_t)vk;
        }
    }
    return true;
}

// ---- Eight-wide compressed packets + their triangle records (kernel variant 14; pt_kernels.h: CW8Packet).
// Built from the reference's binary tree (`src`: the uploaded 48-byte records, every internal box containing its children's -- the
// caller has checked -- and every coordinate finite): a packet stands for a binary node and holds up to eight of its descendants,
// obtained by opening the internal one with the largest surface area until eight are in hand (the 4-ary collapse, carried on).
// The children then take SLOTS by where they lie: slot s points towards (s & 1 ? +x : -x, s & 2 ? +y : -y, s & 4 ? +z : -z), and
// pairs (child, slot) are matched greedily by the projection of the child's centre, relative to the node's, onto the slot's
// direction -- so that a ray whose direction signs form the octant o meets the slots in descending `s ^ o` roughly front to back.
// Internal children are numbered consecutively in ascending slot order (breadth-first queue); a node's leaf children get the
// records base + slot of a record array of their own (slots between its first and last leaf slot are allocated; a slot in
// between that holds no leaf stays an inert record).  Weights: 8 bits against a per-node power of two, rounded up.
bool build_cw8(const uint8_t *src, size_t n, const float *verts /* 12 floats per triangle: a, pad, b, pad, c, pad */, size_t nt,
                      const std::vector<float> &wmax, Cw8Build &out, bool greedy)
{
    WideCollapse plan;
    std::vector<int32_t> entries;
    if (!greedy && !collapse_optimal(src, n, 8, nullptr, plan)) return false;
    auto is_leaf = [&](size_t i) { return ldi(src + i * MI3PT_BVHNODE_STRIDE, 28) == 1; };
    auto box = [&](size_t i, int k) { return ldf(src + i * MI3PT_BVHNODE_STRIDE, (size_t)(k < 3 ? 4 * k : 16 + 4 * (k - 3))); };
    auto area = [&](size_t i) {
        const double x = (double)box(i, 3) - box(i, 0), y = (double)box(i, 4) - box(i, 1), z = (double)box(i, 5) - box(i, 2);
        const double a = x * y + x * z + y * z;
        return a == a ? a : 0.0;
    };
    if (n == 0 || is_leaf(0)) return false;
    const float inf = __builtin_inff();
    std::vector<uint32_t> queue, depth;         // binary node a packet stands for; its level
    std::vector<std::array<int32_t, 8>> kids;   // per packet: binary node per SLOT (-1: empty)
    std::vector<uint32_t> child_base;
    queue.push_back(0); depth.push_back(1);
    int height = 1;
    size_t total_children = 0;
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const uint32_t x = queue[qi];
        const uint8_t *r = src + (size_t)x * MI3PT_BVHNODE_STRIDE;
        int32_t set[8] = { ldi(r, 32), ldi(r, 36), -1, -1, -1, -1, -1, -1 };
        if (set[0] < 0 || set[1] < 0 || (size_t)set[0] >= n || (size_t)set[1] >= n) return false;
        int cnt = 2;
        if (greedy) {
            while (cnt < 8) {
                int pick = -1;
                double best_area = -1.0;
                for (int k = 0; k < cnt; k++) {
                    if (is_leaf((size_t)set[k])) continue;
                    const double a = area((size_t)set[k]);
                    if (a > best_area) { best_area = a; pick = k; }
                }
                if (pick < 0) break;
                const uint8_t *cr = src + (size_t)set[pick] * MI3PT_BVHNODE_STRIDE;
                const int32_t cl = ldi(cr, 32), crr = ldi(cr, 36);
                if (cl < 0 || crr < 0 || (size_t)cl >= n || (size_t)crr >= n) return false;
                set[pick] = cl;
                set[cnt++] = crr;
            }
        } else {
            plan.children_of(x, entries);
            if (entries.size() < 2 || entries.size() > 8) return false;
            cnt = (int)entries.size();
            for (int k = 0; k < cnt; k++) set[k] = entries[(size_t)k];
        }
        total_children += (size_t)cnt;
        // slots: greedy matching of (child, slot) by the projection of the child's centre offset on the slot's direction
        double off[8][3];
        for (int k = 0; k < cnt; k++)
            for (int a = 0; a < 3; a++)
                off[k][a] = 0.5 * ((double)box((size_t)set[k], a) + box((size_t)set[k], 3 + a)) - 0.5 * ((double)box(x, a) + box(x, 3 + a));
        std::array<int32_t, 8> slots = { -1, -1, -1, -1, -1, -1, -1, -1 };
        bool child_done[8] = { false, false, false, false, false, false, false, false };
        for (int round = 0; round < cnt; round++) {
            int bk = -1, bs = -1;
            double bscore = -1e300;
            for (int k = 0; k < cnt; k++) {
                if (child_done[k]) continue;
                for (int sl = 0; sl < 8; sl++) {
                    if (slots[(size_t)sl] >= 0) continue;
                    const double sc = (sl & 1 ? off[k][0] : -off[k][0]) + (sl & 2 ? off[k][1] : -off[k][1]) + (sl & 4 ? off[k][2] : -off[k][2]);
                    if (sc > bscore) { bscore = sc; bk = k; bs = sl; }
                }
            }
            if (bk < 0) return false;
            slots[(size_t)bs] = set[bk];
            child_done[bk] = true;
        }
        kids.push_back(slots);
        child_base.push_back((uint32_t)queue.size());
        for (int sl = 0; sl < 8; sl++) {
            const int32_t c = slots[(size_t)sl];
            if (c >= 0 && !is_leaf((size_t)c)) {
                queue.push_back((uint32_t)c);
                depth.push_back(depth[qi] + 1);
                if ((int)depth[qi] + 1 > height) height = (int)depth[qi] + 1;
            }
        }
        if (queue.size() >= (1u << 24)) return false;
    }
    const size_t np = kids.size();
    out.packets.assign(np, pt::CW8Packet());
    std::memset(out.packets.data(), 0, np * sizeof(pt::CW8Packet));
    std::vector<uint32_t> rec_base(np, 0);
    size_t next_rec = 8;
    for (size_t w = 0; w < np; w++) {
        int first = -1, last = -1;
        for (int sl = 0; sl < 8; sl++)
            if (kids[w][(size_t)sl] >= 0 && is_leaf((size_t)kids[w][(size_t)sl])) { if (first < 0) first = sl; last = sl; }
        if (first >= 0) { rec_base[w] = (uint32_t)(next_rec - (size_t)first); next_rec += (size_t)(last - first + 1); }
    }
    if (next_rec + 8 >= (1u << 24)) return false;
    out.records.assign(next_rec, pt::TriPacket64());
    for (auto &q : out.records) {      // inert: an empty box, a degenerate triangle
        for (int k = 0; k < 3; k++) { q.a[k] = q.e1[k] = q.e2[k] = 0.0f; q.bmin[k] = 1.0f; q.bmax[k] = -1.0f; }
        q.unsafe = 0;
    }
    for (size_t w = 0; w < np; w++) {
        pt::CW8Packet &c = out.packets[w];
        const auto &ks = kids[w];
        uint32_t imask = 0, nchild = 0;
        for (int sl = 0; sl < 8; sl++)
            if (ks[(size_t)sl] >= 0) { nchild++; if (!is_leaf((size_t)ks[(size_t)sl])) imask |= 1u << sl; }
        uint32_t meta = imask << 24;
        for (int ax = 0; ax < 3; ax++) {
            double lo = 1e300, hi = -1e300, maxabs = 0.0;
            for (int sl = 0; sl < 8; sl++) {
                if (ks[(size_t)sl] < 0) continue;
                const size_t ci = (size_t)ks[(size_t)sl];
                lo = std::min(lo, (double)box(ci, ax)); hi = std::max(hi, (double)box(ci, 3 + ax));
                maxabs = std::max({ maxabs, std::fabs((double)box(ci, ax)), std::fabs((double)box(ci, 3 + ax)) });
            }
            if (!(lo <= hi)) return false;
            if (!(maxabs < 1e30)) return false;
            // (the grid of CWidePacket: the extent in at most 248 cells, no finer than 2^-20 of the largest coordinate)
            int e = -100;
            if (hi > lo) e = std::max(e, (int)std::ceil(std::log2((hi - lo) / 248.0)));
            if (maxabs > 0.0) e = std::max(e, (int)std::floor(std::log2(maxabs)) - 20);
            double cell = std::ldexp(1.0, e);
            float o = 0.0f;
            for (;; e++, cell *= 2.0) {
                o = (float)(lo - 2.0 * cell);
                if ((double)o > lo - cell) continue;
                if (std::ceil((hi - (double)o) / cell) + 1.0 <= 254.0) break;
            }
            if (e + 127 < 1 || e + 127 > 254) return false;
            c.o[ax] = o;
            meta |= (uint32_t)(e + 127) << (8 * ax);
            const float cf = (float)cell;
            for (int sl = 0; sl < 8; sl++) {
                uint32_t a = 255u, z = 0u;
                if (ks[(size_t)sl] >= 0) {
                    const size_t ci = (size_t)ks[(size_t)sl];
                    const float b0 = box(ci, ax), b1 = box(ci, 3 + ax);
                    const double x0 = ((double)b0 - (double)o) / cell, x1 = ((double)b1 - (double)o) / cell;
                    const double f0 = std::floor(x0) - 1.0, f1 = std::ceil(x1) + 1.0;
                    if (!(f0 >= 0.0 && f1 <= 254.0 && f0 < f1)) return false;
                    a = (uint32_t)f0; z = (uint32_t)f1;
                    if (!(std::fma((float)a, cf, o) <= b0 && std::fma((float)z, cf, o) >= b1)) return false;      // (the plain-division path's decode: see CWidePacket's builder)
                }
                c.qlo[ax][sl >> 2] |= a << (8 * (sl & 3));
                c.qhi[ax][sl >> 2] |= z << (8 * (sl & 3));
            }
        }
        c.meta = meta;
        // culling weights: wq_k * 2^(wexp - 127) >= W_k; a child that must never be skipped (W = +inf) makes the whole node
        // never skip (wexp 255: the scale is +inf, every product +inf or NaN, neither of which skips)
        double wm = 0.0;
        bool never = false;
        for (int sl = 0; sl < 8; sl++) {
            if (ks[(size_t)sl] < 0) continue;
            const float wv = wmax[(size_t)ks[(size_t)sl]];
            if (!(wv < inf) || wv < 0.0f) never = true; else wm = std::max(wm, (double)wv);
        }
        uint32_t wexp = 255;
        if (!never) {
            int e = wm > 0.0 ? (int)std::ceil(std::log2(wm / 255.0)) : -63;
            if (e < -63) e = -63;                  // (no smaller: the kernel's product rc * 2^e must stay far from the denormals)
            while (std::ceil(wm / std::ldexp(1.0, e)) > 255.0) e++;
            if (e + 127 > 254) never = true; else wexp = (uint32_t)(e + 127);
        }
        if (never) wexp = 255;
        for (int sl = 0; sl < 8; sl++) {
            uint32_t q = 0;
            if (ks[(size_t)sl] >= 0) {
                if (never) q = 1;
                else {
                    const double sc = std::ldexp(1.0, (int)wexp - 127);
                    q = (uint32_t)std::ceil((double)wmax[(size_t)ks[(size_t)sl]] / sc);
                    if (q > 255u) return false;
                    if ((double)q * sc < (double)wmax[(size_t)ks[(size_t)sl]]) return false;
                }
            }
            c.wq[sl >> 2] |= q << (8 * (sl & 3));
        }
        c.child = (child_base[w] & 0xffffffu) | (wexp << 24);
        c.tri = (rec_base[w] & 0xffffffu) | (nchild << 24);
        for (int sl = 0; sl < 8; sl++) {
            if (ks[(size_t)sl] < 0 || !is_leaf((size_t)ks[(size_t)sl])) continue;
            const size_t li = (size_t)ks[(size_t)sl];
            const uint8_t *lr = src + li * MI3PT_BVHNODE_STRIDE;
            const int32_t ti = ldi(lr, 40);
            if (ti < 0 || (size_t)ti >= nt) return false;
            const float *v = verts + (size_t)ti * 12;
            pt::TriPacket64 &q = out.records[(size_t)rec_base[w] + (size_t)sl];
            for (int k = 0; k < 3; k++) {
                volatile float e1 = v[4 + k] - v[k], e2 = v[8 + k] - v[k];        // one fp32 rounding each (see tri_packet_of)
                q.a[k] = v[k]; q.e1[k] = e1; q.e2[k] = e2;
                q.bmin[k] = ldf(lr, 4 * (size_t)k); q.bmax[k] = ldf(lr, 16 + 4 * (size_t)k);
            }
            q.unsafe = (node_box_safe(src, li) ? 0u : 0x80000000u) | (uint32_t)ti;
        }
    }
    out.height = height;
    out.mean_children = np ? (double)total_children / (double)np : 0.0;
    return true;
}

// Host-only check of the 8-wide packets (no device: `-m "not gpu"` tests call it): builds them for a tree + triangles in the reference's
// layouts exactly as prepare_cull does (weights aside: all zero) and walks the result INDEPENDENTLY of the builder's bookkeeping -- every
// leaf triangle of the tree reachable exactly once, every packet referenced exactly once, a leaf slot's record carrying that triangle's index
// and its leaf's box bit for bit, every decoded child box (the fma the kernel's plain-division path uses) containing everything below it.
// out[0..5] = packets, records, packet levels, children per packet x 1000, leaves reached, 1 if the kernel would be offered these packets.
extern "C" int mi3pt_host_eight_wide_check(const void *nodes, size_t nodes_bytes, const void *triangles, size_t triangles_bytes, int greedy, uint64_t out[6])
{
    if (!nodes || !triangles || !out || nodes_bytes % MI3PT_BVHNODE_STRIDE || triangles_bytes % MI3PT_TRIANGLE_STRIDE)
        return pt_set_error(MI3PT_ERR_INVALID, "mi3pt_host_eight_wide_check: bad argument");
    const size_t n = nodes_bytes / MI3PT_BVHNODE_STRIDE, nt = triangles_bytes / MI3PT_TRIANGLE_STRIDE;
    const uint8_t *src = static_cast<const uint8_t *>(nodes);
    std::vector<float> verts(nt * 12, 0.0f);
    for (size_t t = 0; t < nt; t++)
        for (int v = 0; v < 3; v++) std::memcpy(&verts[t * 12 + 4 * (size_t)v], static_cast<const uint8_t *>(triangles) + t * MI3PT_TRIANGLE_STRIDE + 16 * (size_t)v, 12);
    std::vector<float> wmax(n, 0.0f);
    Cw8Build b;
    for (int k = 0; k < 6; k++) out[k] = 0;
    if (!build_cw8(src, n, verts.data(), nt, wmax, b, greedy != 0)) return pt_set_error(MI3PT_ERR_STATE, "mi3pt_host_eight_wide_check: the tree does not admit the 8-wide packets");
    auto fail = [](const std::string &what) { return pt_set_error(MI3PT_ERR_STATE, "mi3pt_host_eight_wide_check: " + what); };
    // leaf of every triangle in the source tree
    std::vector<int64_t> leaf_of(nt, -1);
    size_t nleaves = 0;
    for (size_t i = 0; i < n; i++)
        if (ldi(src + i * MI3PT_BVHNODE_STRIDE, 28) == 1) {
            const int32_t ti = ldi(src + i * MI3PT_BVHNODE_STRIDE, 40);
            if (ti < 0 || (size_t)ti >= nt || leaf_of[(size_t)ti] >= 0) return fail("the source tree is not proper");
            leaf_of[(size_t)ti] = (int64_t)i;
            nleaves++;
        }
    const size_t np = b.packets.size();
    std::vector<uint8_t> pseen(np, 0), tseen(nt, 0);
    std::vector<std::array<float, 6>> below(np);      // union of the LEAF boxes below each packet (filled bottom-up: children have larger indices)
    size_t reached = 0;
    int levels = 0;
    std::vector<int> level(np, 0);
    level[0] = 1; pseen[0] = 1;
    for (size_t w = 0; w < np; w++) {                 // parents before children: propagate levels, check references
        const pt::CW8Packet &c = b.packets[w];
        if (!pseen[w]) return fail("a packet nobody refers to");
        const uint32_t imask = c.meta >> 24, base = c.child & 0xffffffu;
        uint32_t rank = 0;
        for (int sl = 0; sl < 8; sl++)
            if (imask & (1u << sl)) {
                const size_t ch = (size_t)base + rank++;
                if (ch >= np || ch <= w || pseen[ch]) return fail("a child reference out of range, not after its parent, or shared");
                pseen[ch] = 1;
                level[ch] = level[w] + 1;
            }
        if (level[w] > levels) levels = level[w];
    }
    for (size_t w = np; w-- > 0;) {                   // children before parents: boxes
        const pt::CW8Packet &c = b.packets[w];
        const uint32_t imask = c.meta >> 24, base = c.child & 0xffffffu, rbase = c.tri & 0xffffffu;
        float cell[3];
        for (int ax = 0; ax < 3; ax++) { const uint32_t e = (c.meta >> (8 * ax)) & 0xffu; const uint32_t bits = e << 23; std::memcpy(&cell[ax], &bits, 4); }
        std::array<float, 6> u = { 1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f };
        uint32_t rank = 0, children = 0;
        for (int sl = 0; sl < 8; sl++) {
            uint32_t qa[3], qz[3];
            bool empty = true;
            for (int ax = 0; ax < 3; ax++) {
                qa[ax] = (c.qlo[ax][sl >> 2] >> (8 * (sl & 3))) & 0xffu; qz[ax] = (c.qhi[ax][sl >> 2] >> (8 * (sl & 3))) & 0xffu;
                if (!(qa[ax] == 255u && qz[ax] == 0u)) empty = false;
            }
            const bool internal = (imask >> sl) & 1u;
            if (empty) { if (internal) return fail("an empty slot marked internal"); continue; }
            children++;
            std::array<float, 6> cb;                   // what lies below this slot
            if (internal) cb = below[(size_t)base + rank++];
            else {
                const pt::TriPacket64 &q = b.records[(size_t)rbase + (size_t)sl];
                const uint32_t ti = q.unsafe & 0x7fffffffu;
                if (ti >= nt || leaf_of[ti] < 0 || tseen[ti]) return fail("a record without a leaf of its own");
                tseen[ti] = 1; reached++;
                const uint8_t *lr = src + (size_t)leaf_of[ti] * MI3PT_BVHNODE_STRIDE;
                for (int k = 0; k < 3; k++) {
                    if (std::memcmp(&q.bmin[k], lr + 4 * (size_t)k, 4) || std::memcmp(&q.bmax[k], lr + 16 + 4 * (size_t)k, 4)) return fail("a record whose box is not its leaf's box");
                    const float a = verts[(size_t)ti * 12 + (size_t)k];
                    if (std::memcmp(&q.a[k], &a, 4)) return fail("a record whose vertex is not its triangle's");
                    cb[(size_t)k] = q.bmin[k]; cb[3 + (size_t)k] = q.bmax[k];
                }
            }
            for (int ax = 0; ax < 3; ax++) {
                const float lo = std::fma((float)qa[ax], cell[ax], c.o[ax]), hi = std::fma((float)qz[ax], cell[ax], c.o[ax]);
                if (!(lo <= cb[(size_t)ax] && hi >= cb[3 + (size_t)ax])) return fail("a decoded box that does not contain what is below it");
                u[(size_t)ax] = std::min(u[(size_t)ax], cb[(size_t)ax]); u[3 + (size_t)ax] = std::max(u[3 + (size_t)ax], cb[3 + (size_t)ax]);
            }
        }
        if (children != ((c.tri >> 24) & 15u) || children < 2) return fail("a packet's child count");
        below[w] = u;
    }
    if (reached != nleaves) return fail("not every leaf is reachable");
    if (levels != b.height) return fail("the builder's height is not the tree's");
    out[0] = np; out[1] = b.records.size(); out[2] = (uint64_t)b.height; out[3] = (uint64_t)(b.mean_children * 1000.0 + 0.5); out[4] = reached;
    out[5] = b.height <= pt::SM_W8_MIN_LDS_NODES + pt::SM_W8_OVERFLOW_NODES ? 1 : 0;
    return MI3PT_OK;
}

