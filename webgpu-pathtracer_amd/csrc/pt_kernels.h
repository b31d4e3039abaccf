// pt_kernels.h -- parameter blocks shared by the kernel file and the context.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pt {

// Upper bound on the persistent kernels' grid (CUs x waves per CU, clamped to this); sizes the
// per-wave overflow-stack slices and diagnostic slots.  MI355X: 256 CUs x at most 24 waves.
#ifndef PT_MAX_WAVES_PER_CU
#define PT_MAX_WAVES_PER_CU 24
#endif
constexpr int PT_MAX_RESIDENT_WAVES = 256 * PT_MAX_WAVES_PER_CU;

// Stack entries per lane that the state-machine kernels keep in LDS ([depth][lane], 256 B per entry and wave).  With the
// 1.5 KB of parked path state, 20 one-wave workgroups per CU x (SM_LDS_DEPTH x 256 B + 1536 B) must fit the 160 KB of LDS --
// and not fill it to the last byte: with 26 entries (exactly 163 840 B) the twentieth wave was not resident and the
// persistent grid ran with a straggler (dragon-class 15.4 instead of 16.0 Grays/s, profiles/r03_b_occupancy_ab.log).
#ifndef PT_SM_LDS_DEPTH_VALUE
#define PT_SM_LDS_DEPTH_VALUE 24
#endif
constexpr int SM_LDS_DEPTH = PT_SM_LDS_DEPTH_VALUE;
// ... and of the SIX-waves-per-SIMD builds (round 5: the compressed-wide walk of ordinary trees): 24 one-wave workgroups per CU leave
// 6 826 B each, LDS is handed out in 1 280-byte granules, five of them = 6 400 B = 19 entries + the 1.5 KB of parked path state.
#ifndef PT_SM_LDS_DEPTH_SIX_VALUE
#define PT_SM_LDS_DEPTH_SIX_VALUE 19
#endif
constexpr int SM_LDS_DEPTH_SIX = PT_SM_LDS_DEPTH_SIX_VALUE;
constexpr int SM_LDS_DEPTH_SIX_DEEP = 25;       // ... of the six-wave build for very large trees: all 6 400 B are stack, the parked path state lives in memory (RtLaunch::park)
// Stack entries per lane beyond the LDS part: a wave's slice of the global overflow area (the reference aborts a walk at 64
// stacked entries, raytrace.wgsl:167-171: PT_MAX_STACK in pt_kernels.hip), sized for the build with the shallowest LDS part
constexpr int SM_OVERFLOW_ENTRIES = 64 - (SM_LDS_DEPTH_SIX < SM_LDS_DEPTH ? SM_LDS_DEPTH_SIX : SM_LDS_DEPTH);
// (the 8-wide walk, variant 14, keeps 64-bit node entries -- two dwords each -- in the same column and the same overflow slice: SM_W8_OVERFLOW_NODES below)
static_assert(SM_LDS_DEPTH_SIX_DEEP >= SM_LDS_DEPTH_SIX && 4 * 6 * SM_LDS_DEPTH_SIX_DEEP * 256 < 160 * 1024, "six-wave deep build: LDS");
// The culling walks (CULL, WIDE) visit children near first, so their stack occupancy is not the reference order's.
// They keep a fixed leaf list of SM_CULL_LEAF_CAP entries at the top of the LDS column, node entries in the
// SM_LDS_DEPTH - SM_CULL_LEAF_CAP slots below it, and deeper node entries (rare) in the wave's global overflow slice;
// the context offers them only when the order-independent worst case (every box hit, every child possibly first) fits
// both together: SM_CULL_STACK_MAX entries.
#ifndef PT_CULL_LEAF_CAP_VALUE
#define PT_CULL_LEAF_CAP_VALUE 8
#endif
constexpr int SM_CULL_LEAF_CAP = PT_CULL_LEAF_CAP_VALUE;
constexpr int SM_CULL_STACK_MAX = 64 - SM_CULL_LEAF_CAP;       // (LDS node slots + overflow entries of any build: depth - leaf cap + 64 - depth)
// The WIDE walk parks up to four leaves per node step (its `full` rule needs LCAP - 4 >= 0 free slots to ever run one), and
// the culling walks pop one entry and push up to three more than they popped below the leaf list.
static_assert(SM_CULL_LEAF_CAP >= 4 && SM_LDS_DEPTH - SM_CULL_LEAF_CAP >= 3 && SM_LDS_DEPTH_SIX - SM_CULL_LEAF_CAP >= 3, "leaf list / node slots of the culling walks");
static_assert(SM_LDS_DEPTH >= 8 && SM_LDS_DEPTH <= 32 && SM_LDS_DEPTH_SIX >= 8 && SM_LDS_DEPTH_SIX <= 32, "LDS stack depth");

// Resident waves per SIMD the state-machine kernels are compiled for (__launch_bounds__: 5 -> at most 96 vector registers,
// 4 -> 128) and, times four SIMDs, the one-wave workgroups per compute unit of their persistent grid: the tuned twins of
// the shipped walks need 95-96 registers since their service step's scalars moved to memory (RtService) and run five per
// SIMD; the diagnostic twins and the other variants keep four.
#ifndef PT_SM_TUNED_WAVES
#define PT_SM_TUNED_WAVES 5
#endif
constexpr int SM_TUNED_WAVES_PER_SIMD = PT_SM_TUNED_WAVES, SM_OTHER_WAVES_PER_SIMD = 4;
// Round 5: the compressed-wide walk's builds for ordinary trees (walk_min 32) fit 80 registers -- 8 of them spilled around the walk
// loop, service-step state that the walk does not touch -- and run SIX waves per SIMD on a 19-entry LDS stack: dragon +2.8 %, demo
// +3.9 %, close-up +5.2 % (profiles/r05_f_ab_six_waves.log); the deep walks of very large trees (walk_min 44) lose 2.5 % to the shorter
// stack -- their six-wave build therefore keeps the parked path state in memory and gives all 6 400 B to the stack: 25 entries, +4.0 %.
#ifndef PT_SM_SIX_WAVES
#define PT_SM_SIX_WAVES 6
#endif
constexpr int SM_SIX_WAVES_PER_SIMD = PT_SM_SIX_WAVES;
static_assert(4 * SM_TUNED_WAVES_PER_SIMD * (SM_LDS_DEPTH * 256 + 6 * 256) < 160 * 1024, "LDS: stack + parked path state of every resident wave");
static_assert(4 * SM_SIX_WAVES_PER_SIMD * (SM_LDS_DEPTH_SIX * 256 + 6 * 256) < 160 * 1024, "LDS: stack + parked path state of every resident wave (six-wave builds)");

// The environment texture's size: fixed by the API (renderer.ts:76-85; MI3PT_ENV_WIDTH / _HEIGHT in include/mi3pt.h, tied to these
// by a static_assert in pt_context.hip).  The tuned kernel instantiations have it as a constant; launch_raytrace sends any other size
// to the generic twin.
constexpr int ENV_W = 1024, ENV_H = 512;

// per-pass counters, see mi3pt_counter in include/mi3pt.h
enum { CNT_RAYS, CNT_BOX, CNT_TRI, CNT_HIT, CNT_MISS, CNT_OVERFLOW, CNT_PIXELS, CNT_RESERVED, CNT_COUNT };

// child references carried by node packets and by the packet walk's stack
constexpr uint32_t REF_LEAF = 0x80000000u;   // | triangleIndex
constexpr uint32_t REF_NONE = 0xffffffffu;   // absent child (left/right < 0)

// "BVH node packet": one 64-byte record per INTERNAL node holding both children's
// boxes and what is needed to continue below them, so that one aligned 64-B read
// replaces the reference walk's three 48-B record reads per internal node
// (raytrace.wgsl:175, 185, 193).  Values are bit copies of the uploaded records.
struct NodePacket {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t lref, rref;     // child reference: leaf -> 0x80000000 | triangleIndex, else packet index
    uint32_t flags;          // bit0 / bit1: left / right box has a non-zero coordinate outside [2^-70, 2^60]
    uint32_t cull;           // CULL walk: upper bounds of W = |e1|*|e2|*c1 over the triangles below the left (high
                             // 16 bits) / right (low 16 bits) child, as the top halves of binary32 values rounded
                             // up; 0x7f80 = +infinity = never skip.  Written by the context's cull analysis.
};
static_assert(sizeof(NodePacket) == 64, "packet is one 64-B line");

// "Wide packet": up to FOUR child boxes of a node obtained by absorbing internal children into their
// parent (a child is absorbed only if its box contains its own children's boxes, and boxes are bit
// copies of the uploaded records).  The fp32 slab test is monotone under nesting (every operation in
// it is a correctly rounded, hence monotone, function of the box coordinates), so a ray that the
// reference's test rejects at an absorbed node is rejected at each of that node's children too: the
// wide walk reaches exactly the leaves the reference's walk reaches, in half the node steps.
// 128 bytes = two lines; used by the WIDE walk (kernel variant 10).
struct WidePacket {
    float b01[12];           // boxes 0 and 1 in the NodePacket arrangement: mn0 mx0 mn1 mx1
    float b23[12];           // boxes 2 and 3
    uint32_t ref[4];         // child references (leaf: REF_LEAF | triangle, else wide packet index, REF_NONE: empty slot)
    uint32_t cull01, cull23; // culling weights, 16 bits each (as NodePacket::cull)
    uint32_t flags;          // bits 0-3: box i has a non-zero coordinate outside [2^-70, 2^60]; bits 4-6: number of children
    uint32_t pad;
};
static_assert(sizeof(WidePacket) == 128, "two 64-B lines");

// "Compressed wide packet" (kernel variant 13; round-3 verdict item 4): the four child boxes of a wide packet quantised to 8 bits
// per coordinate on a per-node grid -- origin o (fp32) and one power-of-two cell size 2^e per axis -- and rounded OUTWARD by at
// least one whole cell: decoded box k = [o + 2^e (qlo_k), o + 2^e (qhi_k)] contains child k's box with a margin of >= one cell
// on every side.  64 bytes = one half line: four 16-byte loads per node step instead of eight.  Internal boxes need not be exact:
// the reference reaches a leaf iff the exact slab test passes for the LEAF's box (all boxes nested: the test is monotone under
// nesting, every ancestor then passes too), so any test that never rejects a box the exact test would pass is enough above the
// leaves (pt_kernels.hip: cwide_test), and the leaf's own box is tested exactly in the triangle step (TriPacket64).
struct CWidePacket {
    float o[3];              // grid origin: at least two cells below the smallest child coordinate on each axis
    uint32_t meta;           // bits 0-7 / 8-15 / 16-23: biased exponents of the cell size per axis (cell = 2^(e - 127)); bits 24-26: children
    uint32_t qlo[3];         // per axis: the four children's lower cell indices, one byte each (child k = byte k); empty slot: 255
    uint32_t qhi[3];         // ... upper cell indices; empty slot: 0
    uint32_t cull01, cull23; // culling weights, 16 bits each (as WidePacket)
    uint32_t ref[4];         // child references (as WidePacket)
};
static_assert(sizeof(CWidePacket) == 64, "one 64-B half line");

// 64-byte triangle record of the compressed-wide walk: vertex a, the edges (as TriPacket), and the LEAF's box as uploaded -- the
// exact slab test the reference runs before it pushes the leaf (raytrace.wgsl:184-198) happens in the triangle step, from here.
struct TriPacket64 {
    float a[3], e1[3], e2[3];
    float bmin[3], bmax[3];
    uint32_t unsafe;         // 1: the leaf's box has a non-zero coordinate outside [2^-70, 2^60] (its exact test takes the plain divisions)
};
static_assert(sizeof(TriPacket64) == 64, "four 16-B vectors");

// "Eight-wide compressed packet" (kernel variant 14, round 6): up to EIGHT children of a node of the 8-ary collapse, boxes on the
// node's 8-bit grid rounded outward exactly as in CWidePacket.  What is new is how a walk continues below it: the children sit in
// SLOTS 0..7 chosen by where they lie in the node (slot bit 0 / 1 / 2 set: towards +x / +y / +z), so that `slot ^ octant` of a
// ray's direction signs is a front-to-back visiting order without any sort; the internal children are numbered consecutively
// (breadth-first, ascending slot): child packet = child base + popcount(internal mask below the slot); and the triangle records
// of the leaf children lie at record base + slot (TriPacket64 records in an array of their own, with the triangle's index).  A node
// step therefore produces two 8-bit HIT MASKS and pushes at most ONE 64-bit node entry {child base; hits in visiting order, internal
// mask} and ONE 32-bit leaf entry {record base, leaf hits} -- where the 4-ary walk sorts four (key, reference) pairs and pushes
// four entries.  80 bytes = five 16-byte loads.  The parity argument is CWidePacket's: above the leaves a test only has to never
// reject what the reference's exact test passes (every box nested), the leaf's own box is tested exactly in the triangle step,
// equal-t ties go by leaf rank; the grouping of the reference tree's nodes into packets is free.
struct CW8Packet {
    float o[3];              // grid origin (as CWidePacket)
    uint32_t meta;           // bits 0-7 / 8-15 / 16-23: biased cell exponents per axis; bits 24-31: mask of the slots that hold INTERNAL children
    uint32_t qlo[3][2];      // per axis: lower cell indices of slots 0-3, 4-7 (one byte each); empty slot: 255
    uint32_t qhi[3][2];      // ... upper cell indices; empty slot: 0
    uint32_t wq[2];          // culling weights of slots 0-3, 4-7: W_k <= wq_k * 2^(wexp - 127), rounded up (empty / leafless: 0)
    uint32_t child;          // bits 0-23: packet index of the first internal child; bits 24-31: wexp (255: never skip below this node)
    uint32_t tri;            // bits 0-23: record base (slot s -> record base + s); bits 24-27: number of children (the box-test count)
};
static_assert(sizeof(CW8Packet) == 80, "five 16-B vectors");
#ifndef PT_W8_LEAF_CAP_VALUE
#define PT_W8_LEAF_CAP_VALUE 3
#endif
constexpr int SM_W8_LEAF_CAP = PT_W8_LEAF_CAP_VALUE;             // 32-bit leaf entries (each: up to eight parked leaves of one node) at the top of the LDS column
constexpr int SM_W8_MIN_LDS_NODES = (SM_LDS_DEPTH_SIX - SM_W8_LEAF_CAP) / 2;      // 64-bit node entries in LDS of the build with the shortest column
constexpr int SM_W8_OVERFLOW_NODES = 22;      // 64-bit node entries per lane in a wave's global overflow slice (2 x 22 <= SM_OVERFLOW_ENTRIES)

// 48-byte triangle record for intersection only: vertex a, the material index, and the two EDGES b - a and c - a as
// Moller-Trumbore forms them first (raytrace.wgsl:82-83) -- one fp32 subtraction each, rounded to nearest, wherever it is
// computed: at upload (pt_context.hip: tri_packet_of, the host's float subtraction) instead of per test (six vector
// instructions of every triangle test).  The vertex normals (only needed for the one closest hit per ray) stay in the
// uploaded 112-B records, like b and c themselves.
struct TriPacket {
    float a[3]; uint32_t material;
    float e1[3]; uint32_t pad0;      // fl(b - a)
    float e2[3]; uint32_t pad1;      // fl(c - a)
};
static_assert(sizeof(TriPacket) == 48, "three 16-B vectors");

struct SceneRefs {
    const float4 *tris;     // reference layout, 7 x float4 per triangle
    const float4 *nodes;    // reference layout, 3 x float4 per node
    const float4 *mats;     // reference layout, 4 x float4 per material
    const float4 *env;      // rgba32float texels
    const float4 *packets;  // NodePacket array (internal nodes, breadth-first), or null
    const float4 *wide;     // WidePacket array (WIDE walk), or null
    const float4 *tripk;    // TriPacket array, or null
    const float4 *cwide;    // CWidePacket array (kernel variant 13: same numbering as `wide`), or null
    const float4 *tripk64;  // TriPacket64 array (kernel variant 13), or null
    const float4 *cw8;      // CW8Packet array (kernel variant 14), or null
    const float4 *tripk8;   // TriPacket64 array of variant 14: record base + slot; `unsafe` = guard flag << 31 | triangle index
    const uint32_t *leaf_rank;  // per triangle: rank of its leaf in the reference's visiting order
    int32_t leaf_cap;       // > 0: leaves may be tested out of order; LDS slots available for deferred leaves
    int32_t wide_leaf_cap;  // > 0: the WIDE walk is offered (wide packets built, stack bound holds)
    uint32_t wide_root;     // reference of node 0 in wide-packet terms
    const float4 *cdf;      // environment CDF texels (R marginal, G conditional, B sin-weighted luminance), or null
    int32_t env_sampling;   // 1: the reference's dormant importance-sampling lines run (per-pixel kernels only)
    uint32_t ntris, nnodes, nmats, npackets;
    uint32_t root_ref;      // reference of node 0 in packet terms
    uint32_t flags;         // bit0: every ROOT box coordinate is 0 or within [2^-70, 2^60]; bit1 (wide packets): the root's box contains the
                            // boxes of its children, so the wide walk may start at the root packet without testing the root's own box;
                            // bit2: the scene's culling margins are negligible (the one-axis culling condition: what `auto` = 12 means)
    float cull_ka, cull_kb; // CULL walk: scene constants of the distance bound (pt_kernels.hip cull_setup; context: prepare_cull)
    int32_t env_w, env_h;
};

// raytrace.wgsl:66-75, decoded from the 96-byte block
struct RtUniforms {
    float res_x, res_y, aspect;
    uint32_t frame;
    int32_t max_bounces, samples_per_frame;
    float cam_pos[3], cam_dir[3];
    float fov, focal_distance, aperture;
    float env_intensity, env_rotation;
};

// accumulate.wgsl:1-5
struct AccUniforms {
    uint32_t res_w, res_h, frame, enabled;
};

// fullscreen.wgsl:14-20
struct FsUniforms {
    float res_x, res_y, aspect, scaling;
    uint32_t denoise, tonemapping;
};

struct Tile {
    int32_t tex_w, tex_h;        // full texture size
    int32_t local_rows;          // rows held by this rank
    int32_t rank, nranks, block_rows;
};

// The step-voting knobs' defaults (the context's initial values; the shipped kernels' tuned instantiation has them as constants)
#ifndef PT_DEFAULT_WALK_MIN
#define PT_DEFAULT_WALK_MIN 32
#endif
#ifndef PT_DEEP_WALK_MIN
#define PT_DEEP_WALK_MIN 44        // walk_min for the deep walks of very large trees (the compressed-wide walk's second instantiation)
#endif
#ifndef PT_DEFAULT_LEAF_MIN
#define PT_DEFAULT_LEAF_MIN 24
#endif
#ifndef PT_DEFAULT_SHADE_SPLIT
#define PT_DEFAULT_SHADE_SPLIT 64
#endif
#ifndef PT_DEFAULT_TAIL_POLICY
#define PT_DEFAULT_TAIL_POLICY 7
#endif
#ifndef PT_DEFAULT_JOB_CHUNK
#define PT_DEFAULT_JOB_CHUNK 4
#endif

struct RtService;      // pt_kernels.hip: the launch-invariant scalars of the state-machine kernel's service step

struct RtLaunch {
    SceneRefs scene;
    RtUniforms un;
    AccUniforms acc;
    Tile tile;
    float4 *radiance;            // per-frame output (unfused): nframes slots of slot_pixels texels
    size_t slot_pixels;          // texels per radiance slot
    int32_t nframes;             // frames covered by this launch (frame, frame+1, ...); fused launches: 1
    float4 *accum;               // running mean
    uint64_t *block_counters;    // [gridDim.x][CNT_COUNT]
    uint32_t *tile_counter;      // work queue head of the persistent kernel (zeroed per launch)
    int32_t job_group;           // state-machine kernel: 0 = jobs frame-major; n = every frame of n consecutive tiles before the next n tiles
    int32_t tri_pair;            // deferred-leaf walks: a triangle step tests two parked triangles of a lane that has two
    int32_t job_reverse;         // state-machine kernel: 1 = the job sequence backwards (the image's top band, usually sky, last)
    int32_t job_chunk;           // state-machine kernel: job tickets per draw from the queue while it is long (>= 1)
    uint32_t *stack_overflow;    // state-machine kernel: [grid][32][64] stack entries beyond the LDS part
    uint64_t *wave_times;        // diagnostic: [grid][16] begin / feed-empty / end (100 MHz) + shader cycles + step statistics, or null
    int32_t diag_lite;           // experiment builds, wave_times bound: 1 = the LEAN build with lane counts only (k_raytrace_sm's LITE) instead of the diagnostic twin
    int32_t store_f16;
    int32_t walk_min;            // state-machine kernel: walk while at least this many lanes are walking
    int32_t leaf_min;            // deferred-leaf walk: run a triangle step once this many lanes have a leaf parked
    int32_t tail_policy;         // drain-phase scheduling, bit0: larger of node / triangle group, bit1: batched service, bit2: no shade split
    int32_t shade_split;         // service step: 0 = serve hit and miss lanes together; n = serve the larger group, the other only with >= n lanes
    uint32_t *drain_flag;        // signal word (or null): receives drain_seq when the last job has been handed out
    uint32_t drain_seq;
    int32_t top_packets;         // node packets to stage in LDS per wave (0..64)
    int32_t waves_per_cu;        // persistent kernels: resident one-wave workgroups per CU
    int32_t num_cus;             // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    uint32_t *tile_cost;         // state-machine kernel (diagnostic twin), or null: [tiles of this rank's frame] what the paths of each 8x8 tile
                                 // cost this launch -- 4 per node popped, 3 per triangle tested, 10 per path segment, about their shares of a
                                 // wave's time -- added up with one atomic per finished path: the cost feedback behind tile_perm
    const uint32_t *tile_perm;   // state-machine kernel, or null: position in the job order -> tile of the frame; the context sorts
                                 // the tiles by measured cost, costliest first, so that a launch's last tickets are its cheapest
                                 // tiles and the drain after the queue has run empty is short (any order renders the same bits)
    const float4 *cam_base;      // state-machine kernel, or null: per texel of this rank's image, cam_pos + dir0 * focalDistance -- the part of
                                 // cameraToRay (raytrace.wgsl:219-238, 446) that depends on the PIXEL only, formed once per camera by
                                 // launch_camera_base instead of once per frame of a batch in the service step (same operations, same bits)
    int32_t six_waves;           // the compressed-wide walk's build: 1 = six waves per SIMD, 0 = five, -1 = by the size of the launch (route_waves_per_simd)
    float *park;                 // state-machine kernel builds that park a path's throughput and collected light in MEMORY instead of LDS (the
                                 // six-wave build for very large trees: its whole LDS share is stack): [grid][6][64] floats, or null
    RtService *service;          // device memory for one RtService block (service_block_bytes()), or null: the tuned twin of the
                                 // state-machine kernel reads its service step's scalars from it (launch_raytrace fills it first)
};
size_t service_block_bytes();

// The kernel launch_raytrace runs for a requested variant.  kind: 0 = per-pixel kernel (variant 1 / 2), 1 = state-machine kernel
// (variant 4, 7, 9 .. 12; experiment builds: 5, 6, 8), 2 = k_raytrace_persistent (variant 3, experiment builds); lean: the build
// without diagnostics (DIAG = false); blocks: its grid.
struct RtRoute { int kind, variant; bool lean; int blocks; int waves; bool ymax; int walk_min; };      // (waves per SIMD the build is compiled for; the one-axis culling condition; the walk threshold: what names the instantiation)
RtRoute raytrace_route(const RtLaunch &L, int variant);
bool raytrace_variant_fuses(int variant);      // can launch_raytrace fold the accumulate pass into this variant's kernel?  (the per-pixel kernels)
void launch_raytrace_setup(const RtLaunch &L, bool fuse_accumulate, int variant, hipStream_t s);    // before launch_raytrace, same stream
void launch_raytrace(const RtLaunch &L, bool fuse_accumulate, int variant, hipStream_t s);
// fills `out` (tile.local_rows x tile.tex_w float4) for RtLaunch::cam_base from L.un / L.tile
void launch_camera_base(const RtLaunch &L, float4 *out, hipStream_t s);
void launch_accumulate_batch(const AccUniforms &acc0, const Tile &tile, const float4 *slots, size_t slot_pixels,
                             int nframes, float4 *accum, int store_f16, hipStream_t s);
void launch_accumulate(const AccUniforms &acc, const Tile &tile, const float4 *input, float4 *accum,
                       int store_f16, hipStream_t s);
// `taps`: fullscreen_taps_bytes() of device memory (the de-noise pass's tap table); `taps_current`: an earlier call on
// this stream has filled it from the same fs.res_x / fs.res_y
size_t fullscreen_taps_bytes();
void launch_fullscreen(const FsUniforms &fs, const float4 *tex, int tex_w, int tex_h, int canvas_w, int canvas_h,
                       void *taps, bool taps_current, float4 *out_f32, uint32_t *out_rgba8, hipStream_t s);
void launch_debug_intersect(const SceneRefs &scene, const float *rays, size_t n, float *out, int variant,
                            hipStream_t s);
void launch_debug_math(int fn, const float *a, const float *b, float *out, size_t n, hipStream_t s);
// Box tests and rays of one counter set ([nblocks][CNT_COUNT]) since the previous call for that set, into host-visible memory:
// out[1] = box tests, out[2] = rays, then (after a system-scope fence) out[0] = seq.  `prev` (2 x u64, device) carries the sums.
void launch_walk_stats(const uint64_t *counters, int nblocks, uint64_t *prev, uint64_t *out, uint64_t seq, hipStream_t s);
int raytrace_grid_blocks(const Tile &tile);
int raytrace_persistent_blocks(const Tile &tile, int nframes, int waves_per_cu, int num_cus, bool tuned = false, int waves_per_simd = 0);
// packs the three position vectors of `ntris` 112-byte triangle records into 48-byte rows (the context's cull analysis)
void launch_pack_vertices(const float4 *tris, float4 *out, uint32_t ntris, hipStream_t s);
// writes NodePacket::cull of `npackets` packets from a dense array (the context's cull analysis)
void launch_patch_cull(float4 *packets, const uint32_t *cull, uint32_t npackets, hipStream_t s);

}  // namespace pt
