// pt_host_scene.cpp -- the reference's CPU-side scene compile, in C++.
//
//   mi3pt_host_build_bvh[_f64]  buildBVH + buildBVHRecursive + flattenBVH
//                               (src/passes/raytrace.ts:540-694)
//   mi3pt_host_env_cdf          the CDF texture of updateEnvironmentTexture
//                               (src/renderer.ts:159-266)
//
// The builder produces the IDENTICAL tree to the reference's O(n^2)-per-node code:
// same axis rule (raytrace.ts:593: x only if x > y, then z unless x > z; y whenever
// x <= y), same stable sort on box centres (:596-600), same full SAH sweep with the
// first strict minimum winning (:626-643), same breadth-first flatten (:667-694).
// The O(n) re-scan per split candidate is replaced by prefix / suffix boxes -- box
// unions are exact min/max operations, so the doubles are the same -- and subtrees are
// built by a small thread pool.  All arithmetic is IEEE double like the JavaScript.
#include "../../include/mi3pt.h"
#include "pt_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Item {            // a leaf BVHNode of buildBVH (raytrace.ts:544-556) or a built subtree
    double mn[3], mx[3];
    int32_t node;        // index into Builder::nodes
};

struct TreeNode {
    double mn[3], mx[3];
    int32_t left, right; // -1 for leaves
    int32_t tri;         // -1 for internal nodes
};

struct Task {
    size_t lo, hi;       // range of items
    int32_t *slot;       // where to store the resulting node index
};

struct Builder {
    std::vector<Item> items;
    std::vector<TreeNode> nodes;
    std::atomic<int32_t> next_node{0};

    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> queue;
    size_t pending = 0;      // tasks queued or running
    bool done = false;

    static constexpr size_t kSpawnThreshold = 4096;

    int32_t alloc_node() { return next_node.fetch_add(1); }

    void push(const Task &t)
    {
        std::lock_guard<std::mutex> g(mu);
        queue.push_back(t);
        pending++;
        cv.notify_one();
    }

    // buildBVHRecursive for items[lo, hi) with n >= 2, splitting once; children that
    // are large go back to the queue, small ones are finished on this thread.
    void build_range(size_t lo, size_t hi, int32_t *slot, std::vector<Task> &local,
                     std::vector<double> &keys, std::vector<uint32_t> &order, std::vector<Item> &tmp,
                     std::vector<double> &suffix)
    {
        const size_t n = hi - lo;
        if (n == 1) {                       // raytrace.ts:567-570
            *slot = items[lo].node;
            return;
        }
        const int32_t id = alloc_node();
        TreeNode &node = nodes[id];
        node.tri = -1;
        for (int k = 0; k < 3; k++) {       // :581-584 expandByPoint(min), expandByPoint(max)
            double mn = std::numeric_limits<double>::infinity(), mx = -mn;
            for (size_t i = lo; i < hi; i++) {
                mn = std::min(mn, std::min(items[i].mn[k], items[i].mx[k]));
                mx = std::max(mx, std::max(items[i].mn[k], items[i].mx[k]));
            }
            node.mn[k] = mn;
            node.mx[k] = mx;
        }
        *slot = id;
        if (n == 2) {                       // :587-589, children as given, no sort
            node.left = items[lo].node;
            node.right = items[lo + 1].node;
            return;
        }
        // :592-593
        const double sx = node.mx[0] - node.mn[0], sy = node.mx[1] - node.mn[1], sz = node.mx[2] - node.mn[2];
        const int axis = sx > sy ? (sx > sz ? 0 : 2) : 1;
        // :596-600 stable sort by box centre (getCenter = (min + max) * 0.5)
        keys.resize(n);
        order.resize(n);
        for (size_t i = 0; i < n; i++) {
            keys[i] = (items[lo + i].mn[axis] + items[lo + i].mx[axis]) * 0.5;
            order[i] = (uint32_t)i;
        }
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
        tmp.resize(n);
        for (size_t i = 0; i < n; i++) tmp[i] = items[lo + order[i]];
        std::copy(tmp.begin(), tmp.end(), items.begin() + (ptrdiff_t)lo);

        // :602-643 SAH sweep; suffix[i] = surface area of the box of items[i, n)
        suffix.resize(n + 1);
        {
            double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (size_t i = n; i-- > 1;) {
                const Item &it = items[lo + i];
                for (int k = 0; k < 3; k++) {
                    mn[k] = std::min(mn[k], std::min(it.mn[k], it.mx[k]));
                    mx[k] = std::max(mx[k], std::max(it.mn[k], it.mx[k]));
                }
                const double x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
                suffix[i] = 2 * (x * y + x * z + y * z);
            }
        }
        double min_cost = std::numeric_limits<double>::infinity();
        size_t min_index = (size_t)-1;
        {
            double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (size_t i = 1; i < n; i++) {
                const Item &it = items[lo + i - 1];
                for (int k = 0; k < 3; k++) {
                    mn[k] = std::min(mn[k], std::min(it.mn[k], it.mx[k]));
                    mx[k] = std::max(mx[k], std::max(it.mn[k], it.mx[k]));
                }
                const double x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
                const double left_area = 2 * (x * y + x * z + y * z);
                const double cost = left_area * (double)i + suffix[i] * (double)(n - i);
                if (cost < min_cost) {
                    min_cost = cost;
                    min_index = i;
                }
            }
        }
        if (min_index == (size_t)-1) min_index = 1;   // all costs NaN/inf: JS slice(0,-1) has no sane
                                                      // meaning; never reached for finite boxes
        // :646-651
        const Task lt = { lo, lo + min_index, &node.left };
        const Task rt = { lo + min_index, hi, &node.right };
        for (const Task &t : { lt, rt }) {
            if (t.hi - t.lo >= kSpawnThreshold) push(t);
            else local.push_back(t);
        }
    }

    void worker()
    {
        std::vector<Task> local;
        std::vector<double> keys, suffix;
        std::vector<uint32_t> order;
        std::vector<Item> tmp;
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return done || !queue.empty(); });
                if (queue.empty()) return;
                t = queue.front();
                queue.pop_front();
            }
            local.clear();
            local.push_back(t);
            while (!local.empty()) {
                const Task cur = local.back();
                local.pop_back();
                build_range(cur.lo, cur.hi, cur.slot, local, keys, order, tmp, suffix);
            }
            {
                std::lock_guard<std::mutex> g(mu);
                pending--;
                if (pending == 0) {
                    done = true;
                    cv.notify_all();
                }
            }
        }
    }
};

}  // namespace

extern "C" int mi3pt_host_build_bvh_f64(const double *positions, size_t ntris, void *nodes_out,
                                        size_t nodes_capacity_bytes, size_t *nnodes_out, int nthreads)
{
    if (!positions || !nodes_out || ntris == 0) {
        // raytrace.ts:563-565 "Input nodes array is empty"
        return pt_set_error(MI3PT_ERR_INVALID, ntris == 0 ? "Input nodes array is empty" : "null argument");
    }
    if (ntris > 0x3fffffffu) return pt_set_error(MI3PT_ERR_INVALID, "too many triangles");
    const size_t nnodes = 2 * ntris - 1;
    if (nodes_capacity_bytes < nnodes * MI3PT_BVHNODE_STRIDE)
        return pt_set_error(MI3PT_ERR_INVALID, "nodes_out too small: need (2*ntris-1)*48 bytes");

    Builder b;
    b.items.resize(ntris);
    b.nodes.resize(nnodes);
    // buildBVH, raytrace.ts:540-556: one leaf per triangle, Box3.setFromPoints
    for (size_t i = 0; i < ntris; i++) {
        const double *p = positions + i * 9;
        TreeNode &leaf = b.nodes[i];
        for (int k = 0; k < 3; k++) {
            leaf.mn[k] = std::min(p[k], std::min(p[3 + k], p[6 + k]));
            leaf.mx[k] = std::max(p[k], std::max(p[3 + k], p[6 + k]));
            b.items[i].mn[k] = leaf.mn[k];
            b.items[i].mx[k] = leaf.mx[k];
        }
        leaf.left = leaf.right = -1;
        leaf.tri = (int32_t)i;
        b.items[i].node = (int32_t)i;
    }
    b.next_node = (int32_t)ntris;

    int32_t root = -1;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    b.push(Task{ 0, ntris, &root });
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) pool.emplace_back([&b] { b.worker(); });
    b.worker();
    for (auto &th : pool) th.join();

    // flattenBVH, raytrace.ts:667-694: breadth-first order, child refs -> indices
    std::vector<int32_t> bfs;
    bfs.reserve(nnodes);
    bfs.push_back(root);
    for (size_t head = 0; head < bfs.size(); head++) {
        const TreeNode &n = b.nodes[bfs[head]];
        if (n.tri < 0) {
            bfs.push_back(n.left);
            bfs.push_back(n.right);
        }
    }
    std::vector<int32_t> index_of(nnodes, -1);
    for (size_t i = 0; i < bfs.size(); i++) index_of[bfs[i]] = (int32_t)i;
    uint8_t *out = static_cast<uint8_t *>(nodes_out);
    std::memset(out, 0, bfs.size() * MI3PT_BVHNODE_STRIDE);
    for (size_t i = 0; i < bfs.size(); i++) {
        const TreeNode &n = b.nodes[bfs[i]];
        uint8_t *rec = out + i * MI3PT_BVHNODE_STRIDE;
        float f[3];
        for (int k = 0; k < 3; k++) f[k] = (float)n.mn[k];
        std::memcpy(rec + 0, f, 12);
        for (int k = 0; k < 3; k++) f[k] = (float)n.mx[k];
        std::memcpy(rec + 16, f, 12);
        const bool leaf = n.tri >= 0;
        const int32_t is_leaf = leaf ? 1 : 0;
        const int32_t left = leaf ? -1 : index_of[n.left];
        const int32_t right = leaf ? -1 : index_of[n.right];
        const int32_t tri = leaf ? n.tri : -1;
        std::memcpy(rec + 28, &is_leaf, 4);
        std::memcpy(rec + 32, &left, 4);
        std::memcpy(rec + 36, &right, 4);
        std::memcpy(rec + 40, &tri, 4);
    }
    if (nnodes_out) *nnodes_out = bfs.size();
    return MI3PT_OK;
}

extern "C" int mi3pt_host_build_bvh(const void *triangles, size_t ntris, void *nodes_out,
                                    size_t nodes_capacity_bytes, size_t *nnodes_out, int nthreads)
{
    if (!triangles || ntris == 0)
        return pt_set_error(MI3PT_ERR_INVALID, ntris == 0 ? "Input nodes array is empty" : "null argument");
    std::vector<double> pos(ntris * 9);
    const uint8_t *t = static_cast<const uint8_t *>(triangles);
    for (size_t i = 0; i < ntris; i++) {
        float v[3];
        for (int c = 0; c < 3; c++) {
            std::memcpy(v, t + i * MI3PT_TRIANGLE_STRIDE + 16 * c, 12);
            for (int k = 0; k < 3; k++) pos[i * 9 + 3 * c + k] = (double)v[k];
        }
    }
    return mi3pt_host_build_bvh_f64(pos.data(), ntris, nodes_out, nodes_capacity_bytes, nnodes_out, nthreads);
}

// renderer.ts:159-266.  JS numbers are doubles; the typed arrays the reference stores
// intermediate results in are Float32Array, so those stores round to fp32.
extern "C" int mi3pt_host_env_cdf(const float *rgba, int width, int height, float *cdf)
{
    if (!rgba || !cdf || width <= 0 || height <= 0) return pt_set_error(MI3PT_ERR_INVALID, "bad argument");
    const size_t W = (size_t)width, H = (size_t)height, N = W * H;
    std::vector<float> luminance(N), weighted(N), marginal(H), conditional(N), rows(H), column(N);
    for (size_t i = 0; i < N; i++) {                       // :163-171
        const double r = rgba[4 * i], g = rgba[4 * i + 1], b = rgba[4 * i + 2];
        luminance[i] = (float)(0.2126 * r + 0.7152 * g + 0.0722 * b);
    }
    for (size_t y = 0; y < H; y++) {                       // :177-187
        const double y_range = ((double)y + 0.5) / (double)H;
        const double theta = y_range * 3.141592653589793;
        const double weight = std::sin(theta);
        for (size_t x = 0; x < W; x++) weighted[y * W + x] = (float)((double)luminance[y * W + x] * weight);
    }
    {                                                      // :191-217
        double total = 0;
        for (size_t y = 0; y < H; y++) {
            double row_total = 0;
            for (size_t x = 0; x < W; x++) row_total += (double)weighted[y * W + x];
            rows[y] = (float)row_total;
            total += row_total;
        }
        for (size_t y = 0; y < H; y++) rows[y] = (float)((double)rows[y] / total);
        double sum = 0;                                     // exclusive prefix == the fresh sum per y
        for (size_t y = 0; y < H; y++) {
            marginal[y] = (float)sum;
            sum += (double)rows[y];
        }
    }
    for (size_t y = 0; y < H; y++) {                       // :223-251
        double row_total = 0;
        for (size_t x = 0; x < W; x++) row_total += (double)luminance[y * W + x];
        for (size_t x = 0; x < W; x++) column[y * W + x] = (float)((double)luminance[y * W + x] / row_total);
        double sum = 0;
        for (size_t x = 0; x < W; x++) {
            conditional[y * W + x] = (float)sum;
            sum += (double)column[y * W + x];
        }
    }
    for (size_t y = 0; y < H; y++)                         // :257-266
        for (size_t x = 0; x < W; x++) {
            const size_t i = y * W + x;
            cdf[4 * i] = marginal[y];
            cdf[4 * i + 1] = conditional[i];
            cdf[4 * i + 2] = weighted[i];
            cdf[4 * i + 3] = 1.0f;
        }
    return MI3PT_OK;
}

extern "C" int mi3pt_tile_local_rows(int height, int rank, int nranks, int block_rows)
{
    if (height < 0 || nranks <= 0 || rank < 0 || rank >= nranks || block_rows <= 0) return -1;
    // rounds of the deal go back and forth: rank r owns block r of even rounds and block nranks - 1 - r of odd ones
    // (include/mi3pt.h, mi3pt_set_tile)
    int n = 0;
    for (int y = 0; y < height; y++) {
        const int gb = y / block_rows, round = gb / nranks, pos = gb % nranks;
        if (((round & 1) ? nranks - 1 - pos : pos) == rank) n++;
    }
    return n;
}

// The deal for hosts that gather on their own (include/mi3pt.h).  Local row ly sits in this rank's block number b = ly / block_rows,
// which it drew in round b: position `rank` of an even round, nranks - 1 - rank of an odd one.
extern "C" int mi3pt_tile_global_row(int local_row, int rank, int nranks, int block_rows)
{
    if (local_row < 0 || nranks <= 0 || rank < 0 || rank >= nranks || block_rows <= 0) return -1;
    const int b = local_row / block_rows;
    const long long y = ((long long)b * nranks + ((b & 1) ? nranks - 1 - rank : rank)) * block_rows + local_row % block_rows;
    return y > 0x7fffffffLL ? -1 : (int)y;
}

extern "C" int mi3pt_tile_owner(int y, int nranks, int block_rows)
{
    if (y < 0 || nranks <= 0 || block_rows <= 0) return -1;
    const int gb = y / block_rows, round = gb / nranks, pos = gb % nranks;
    return (round & 1) ? nranks - 1 - pos : pos;
}
