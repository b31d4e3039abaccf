/*
 * mi3pt.h -- C ABI of libmi3pt.so, the MI355X (gfx950) path-tracing back-end.
 *
 * This is the drop-in boundary for the reference's per-pixel ray-trace path
 * (umar-ahmed/webgpu-pathtracer: src/passes/, src/renderer.ts, src/scene.ts).  The
 * reference has no FFI: its seam is the WebGPU device/queue API as used by the three
 * Pass classes and the Renderer.  Every entry point below replaces one group of those
 * WebGPU calls and cites it (paths relative to the reference root).  Buffers cross
 * the boundary as RAW BYTES IN THE REFERENCE'S OWN LAYOUTS (what webgpu-utils
 * computes from the WGSL structs):
 *
 *   Triangle  112 B  raytrace.wgsl:40-49    BVHNode  48 B  raytrace.wgsl:51-64
 *   Material   64 B  raytrace.wgsl:31-38    Uniforms 96 B  raytrace.wgsl:66-75
 *   accumulate Uniforms 16 B accumulate.wgsl:1-5
 *   fullscreen Uniforms 24 B fullscreen.wgsl:14-20
 *   environment / CDF textures: 1024x512 rgba32float, renderer.ts:111-130
 *
 * Conventions
 *   - plain C types only; every function returns an mi3pt_status (0 = OK) unless
 *     stated otherwise; mi3pt_last_error() gives the message for the calling thread
 *     (the N-API / ctypes shims turn it into a thrown Error, like the reference's
 *     `throw new Error(...)` sites renderer.ts:65-67, 133-143, 514-516).
 *   - uploads COPY at call time (queue.writeBuffer / writeTexture semantics); the
 *     caller keeps ownership of its memory.
 *   - mi3pt_submit() is asynchronous and stream ordered, like
 *     `device.queue.submit([encoder.finish()])` (renderer.ts:389-390); reads and
 *     mi3pt_sync() are the only blocking calls.
 *   - one host thread per context; no callbacks.
 *   - there is NO CPU fallback: without a HIP device mi3pt_create() fails with
 *     MI3PT_ERR_NO_DEVICE.  The mi3pt_host_* functions are the reference's own
 *     CPU-side scene compile (BVH build, env CDF) and need no device.
 */
#ifndef MI3PT_H
#define MI3PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 5): mi3pt_set_tile deals the row blocks back and forth (ABI 2 as first released dealt them one way: a host that
 * de-interleaves gathered rows must use mi3pt_tile_global_row / mi3pt_tile_owner, not its own copy of the old formula);
 * MI3PT_OPT_BATCH defaults to 256 frames per launch (was 64), MI3PT_OPT_WALK_MIN reads 0 = "by the size of the tree" (was 32);
 * new exports mi3pt_tile_global_row, mi3pt_tile_owner; new options MI3PT_OPT_GATE_TIMEOUT_MS, MI3PT_OPT_GATE_RELEASES, MI3PT_OPT_CAMERA_BASE,
 * MI3PT_OPT_PACKET_ORDER, MI3PT_OPT_SIX_WAVES (22 .. 27); MI3PT_OPT_WAVES_PER_CU reads up to 24 (six waves per SIMD). */
/* 4 (round 6): mi3pt_set_rows and mi3pt_measure_tile_cost are GONE (contiguous cost-balanced bands: measured 7 % slower than the dealt
 * 8-row blocks in round 4 and kept since as dead surface); mi3pt_set_kernel_variant accepts 14 (the eight-wide walk: an option). */
#define MI3PT_ABI_VERSION 4

typedef enum mi3pt_status {
    MI3PT_OK = 0,
    MI3PT_ERR_INVALID = 1,    /* bad argument / wrong byte size */
    MI3PT_ERR_NO_DEVICE = 2,  /* no HIP device (renderer.ts:514-516 "WebGPU device not found.") */
    MI3PT_ERR_HIP = 3,        /* a HIP runtime call failed */
    MI3PT_ERR_STATE = 4       /* call made in the wrong state (e.g. submit before resize) */
} mi3pt_status;

/* strides of the reference layouts */
#define MI3PT_TRIANGLE_STRIDE 112
#define MI3PT_BVHNODE_STRIDE 48
#define MI3PT_MATERIAL_STRIDE 64
#define MI3PT_RAYTRACE_UNIFORMS_SIZE 96
#define MI3PT_ACCUMULATE_UNIFORMS_SIZE 16
#define MI3PT_FULLSCREEN_UNIFORMS_SIZE 24
#define MI3PT_ENV_WIDTH 1024
#define MI3PT_ENV_HEIGHT 512

/* the three passes of renderer.ts:23-27 */
typedef enum mi3pt_pass {
    MI3PT_PASS_RAYTRACE = 0,
    MI3PT_PASS_ACCUMULATE = 1,
    MI3PT_PASS_FULLSCREEN = 2
} mi3pt_pass;

#define MI3PT_SUBMIT_RAYTRACE 1u
#define MI3PT_SUBMIT_ACCUMULATE 2u
#define MI3PT_SUBMIT_FULLSCREEN 4u

/* which image mi3pt_read_texture() returns */
typedef enum mi3pt_texture {
    MI3PT_TEX_OUTPUT = 0,        /* renderer.outputTexture (renderer.ts:94-109): the frame's radiance
                                    after a raytrace pass, the running mean after an accumulate pass
                                    (accumulate.ts:171-175 copies it back) */
    MI3PT_TEX_ACCUMULATION = 1,  /* accumulate.ts:45-59 accumulationTexture == outputTexturePrev */
    MI3PT_TEX_CANVAS = 2         /* the fullscreen pass's colour attachment, as float RGBA
                                    (canvas_w x canvas_h, row 0 = top) */
} mi3pt_texture;

/* texel storage of output / accumulation textures */
typedef enum mi3pt_storage {
    MI3PT_STORAGE_F32 = 0,       /* keep fp32 (default; what the 1e-4 parity target needs) */
    MI3PT_STORAGE_F16 = 1        /* round every stored texel through binary16 = the reference's
                                    rgba16float textures (renderer.ts:102, accumulate.ts:52) */
} mi3pt_storage;

/* what a submit that includes MI3PT_SUBMIT_FULLSCREEN shows (mi3pt_set_present_mode) */
typedef enum mi3pt_present_mode {
    MI3PT_PRESENT_EXACT = 0,     /* the canvas is drawn from the running mean INCLUDING this submit's frame:
                                    renderer.ts:379-390 as written, one accumulate pass and one fullscreen pass
                                    per presenting frame, in frame order.  The frames' raytrace passes are
                                    launched together, up to MI3PT_OPT_PRESENT_DEPTH (16) frames at a time,
                                    when the queue is that deep or the context is observed (read-back, sync,
                                    flush): nothing that can be read differs from a launch per frame. */
    MI3PT_PRESENT_LATEST = 1     /* headless hosts: the frame is queued like any other; the canvas is drawn
                                    from the running mean of the batches launched so far, and only when that
                                    mean or the fullscreen uniforms changed since the last draw.  A submit
                                    with FULLSCREEN alone (render() after sampling has stopped,
                                    renderer.ts:369-373) launches the queue and shows every frame. */
} mi3pt_present_mode;

/* counters accumulated by every raytrace pass since the last mi3pt_reset_counters() */
typedef enum mi3pt_counter {
    MI3PT_CNT_RAYS = 0,       /* raySceneIntersect calls (raytrace.wgsl:379) */
    MI3PT_CNT_BOX_TESTS = 1,  /* rayAABBIntersect calls  (raytrace.wgsl:118) = N_box */
    MI3PT_CNT_TRI_TESTS = 2,  /* rayTriangleIntersect calls (raytrace.wgsl:78) = N_tri */
    MI3PT_CNT_HITS = 3,
    MI3PT_CNT_MISSES = 4,     /* environment lookups */
    MI3PT_CNT_STACK_OVERFLOWS = 5, /* traversals aborted at raytrace.wgsl:167-171 */
    MI3PT_CNT_PIXELS = 6,     /* (pixel, frame) jobs executed */
    MI3PT_CNT_RESERVED = 7,
    MI3PT_CNT_COUNT = 8
} mi3pt_counter;

typedef struct mi3pt_ctx mi3pt_ctx;

/* ---- library ---- */
int mi3pt_abi_version(void);
const char *mi3pt_last_error(void);

/* ---- device discovery: Renderer.diagnostic(), renderer.ts:470-489 ---- */
int mi3pt_device_count(int *count);
int mi3pt_device_name(int device, char *name, size_t capacity);

/* ---- context: Renderer.create() + constructor, renderer.ts:47-92, 491-533.
 * Creates the HIP stream, placeholder scene buffers (2 triangles / 1 material /
 * 1 node, raytrace.ts:72-77), zeroed environment textures and default uniforms.
 * destroy waits for submitted work first (renderer.ts:418-429). ---- */
int mi3pt_create(int device, mi3pt_ctx **out_ctx);
int mi3pt_destroy(mi3pt_ctx *ctx);
/* Renderer.create() over several GPUs of one node (the reference knows one adapter and one device, renderer.ts:491-533;
 * SURVEY.md 8e): the returned handle is used with the SAME entry points as a single-device context and renders the
 * same bits.  Behind it: one member context per listed device -- member i renders tile i of n (block_rows-row blocks
 * dealt round robin), every upload is replicated, nothing is exchanged per frame -- and a presenting context on
 * devices[0] that holds the whole image.  Reading the accumulation image or the canvas GATHERS: one strided
 * device-to-device copy per member (peer DMA over xGMI), de-interleaved on the way; no host staging, no collective
 * library.  Differences from a single context, all following from "the image is whole only after the gather":
 *   - mi3pt_read_texture / mi3pt_write_texture move WHOLE images (height x width), whatever the split;
 *   - the fullscreen pass runs on the gathered image, so a group always presents lazily: a FULLSCREEN submit is
 *     remembered and performed when the canvas is read (MI3PT_PRESENT_LATEST whatever mi3pt_set_present_mode says);
 *   - mi3pt_set_tile, mi3pt_set_stream and mi3pt_bind_accumulation are refused (MI3PT_ERR_STATE);
 *   - counters are the members' sums, pass times the slowest member's; the debug probes address member 0.
 * A device may be listed more than once (several members on one GPU: how the tests run it on a one-GPU box). */
int mi3pt_create_group(const int *devices, int ndevices, int block_rows, mi3pt_ctx **out_ctx);
/* Members of a group handle (1 for a plain context) and the member contexts themselves (index -1: the presenting
 * context), e.g. for per-device launch statistics.  They belong to the group: never destroy or resize them. */
int mi3pt_group_size(mi3pt_ctx *ctx, int *members);
int mi3pt_group_member(mi3pt_ctx *ctx, int index, mi3pt_ctx **member);

/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the
 * context's own; NULL restores the internal stream. */
int mi3pt_set_stream(mi3pt_ctx *ctx, void *hip_stream);
int mi3pt_set_storage(mi3pt_ctx *ctx, int storage /* mi3pt_storage */);

/* Tile split (multi-GPU): the image's rows are dealt to the ranks in blocks of block_rows, round after round, BACK AND
 * FORTH: with gb = y / block_rows, round = gb / nranks, pos = gb % nranks this context renders the rows whose
 * (round odd ? nranks - 1 - pos : pos) == rank -- a cost that rises or falls down the image (floor, model, sky) is shared out
 * evenly (dealt one way only, rank 0 of eight had 5 % more work than rank 6).  Its textures are compact local_rows x width images,
 * rows in image order.  Default rank 0 of 1.  Takes effect at the next mi3pt_resize(). */
int mi3pt_set_tile(mi3pt_ctx *ctx, int rank, int nranks, int block_rows);
int mi3pt_tile_local_rows(int height, int rank, int nranks, int block_rows); /* returns the count */
/* The deal itself, for hosts that gather and de-interleave on their own (multi-process jobs: bench.py, a JS host per GPU) -- so
 * that nobody re-implements the formula above: the image row that local row `local_row` of `rank`'s compact texture holds
 * (may be >= height in a ragged last round: such rows do not exist; -1 for bad arguments), and the rank that owns image row y. */
int mi3pt_tile_global_row(int local_row, int rank, int nranks, int block_rows);
int mi3pt_tile_owner(int y, int nranks, int block_rows);

/* ---- scene upload: queue.writeBuffer of the structured views ----
 * raytrace.ts:104-121 (triangles), :138-160 (materials), :177-193 (BVH nodes).
 * nbytes must be a non-zero multiple of the stride.  Buffer handles are resolved at
 * dispatch time, so a new scene is picked up by the next submit (the reference needs
 * a reset() for that: raytrace.ts:403 vs :508-522). */
int mi3pt_upload_triangles(mi3pt_ctx *ctx, const void *bytes, size_t nbytes);
int mi3pt_upload_materials(mi3pt_ctx *ctx, const void *bytes, size_t nbytes);
int mi3pt_upload_bvh(mi3pt_ctx *ctx, const void *bytes, size_t nbytes);

/* renderer.ts:132-157 (environment texels) and :253-280 (CDF texels): width/height
 * must be 1024 x 512 ("Environment texture must be 1024x512 pixels"). */
int mi3pt_upload_environment(mi3pt_ctx *ctx, const float *rgba, int width, int height);
int mi3pt_upload_environment_cdf(mi3pt_ctx *ctx, const float *rgba, int width, int height);

/* ---- textures: resize() / reset(), renderer.ts:283-295, 397-416 and
 * accumulate.ts:36-59.  (Re)allocates and ZEROES the output, accumulation and canvas
 * images; width x height is the canvas / full texture size. ---- */
int mi3pt_resize(mi3pt_ctx *ctx, int width, int height);
int mi3pt_reset(mi3pt_ctx *ctx);

/* ---- uniforms: Pass.setUniforms -> queue.writeBuffer(uniformsBuffer, 0, ...)
 * raytrace.ts:359-369, accumulate.ts:178-188, fullscreen.ts:138-148.  The host keeps
 * the structured view (partial set semantics live there) and sends the whole block:
 * 96 / 16 / 24 bytes. ---- */
int mi3pt_set_uniforms(mi3pt_ctx *ctx, int pass /* mi3pt_pass */, const void *bytes, size_t nbytes);

/* ---- execution: one command buffer, renderer.ts:379-390.  pass_mask is an OR of
 * MI3PT_SUBMIT_*; passes run in the reference's order raytrace -> accumulate ->
 * fullscreen (raytrace.ts:696-708, accumulate.ts:154-176, fullscreen.ts:158-177).
 * RAYTRACE|ACCUMULATE in one submit runs as one fused kernel (bit-identical to
 * the two-pass result). ---- */
int mi3pt_submit(mi3pt_ctx *ctx, unsigned pass_mask);
/* `count` consecutive frames with one call: frame i is submitted with the raytrace and the
 * accumulate uniform `frame` fields at (their current value + i); afterwards both hold current +
 * count.  Exactly what `count` calls of Renderer.render() submit while only the frame counter
 * moves (renderer.ts:369-377); for headless hosts that render a fixed number of samples. */
int mi3pt_submit_frames(mi3pt_ctx *ctx, unsigned pass_mask, uint32_t count);
int mi3pt_sync(mi3pt_ctx *ctx);   /* queue.onSubmittedWorkDone(), renderer.ts:420 */
/* Default MI3PT_PRESENT_EXACT.  See mi3pt_present_mode. */
int mi3pt_set_present_mode(mi3pt_ctx *ctx, int mode /* mi3pt_present_mode */);
/* RAYTRACE|ACCUMULATE submits may be queued inside the library and launched together (up to
 * 256 consecutive frames whose uniforms differ only in `frame` -- 256 x nranks for a rank of a tile
 * split, at most 512 (MI3PT_OPT_BATCH, MI3PT_OPT_BATCH_LIMIT), less when memory is short: a quarter of
 * the free memory bounds the radiance slots -- run as one kernel + one ordered accumulate).  Every call that observes or changes device state launches the queue first;
 * mi3pt_flush does only that, without waiting -- use it before synchronising the stream
 * yourself (e.g. torch.cuda.synchronize()). */
int mi3pt_flush(mi3pt_ctx *ctx);
/* Frames one launch covers at the current size and tile split (the queue is launched by itself when
 * it holds this many).  A caller that wants launches of one shape flushes at a divisor of its job. */
int mi3pt_batch_capacity(mi3pt_ctx *ctx, int *frames);

/* ---- read-back (the capability a headless drop-in needs; the reference only has
 * canvas.toDataURL, main.ts:351-356).  Blocking.  dst holds rows x width x 4 floats
 * (rows = local rows for OUTPUT / ACCUMULATION, canvas height for CANVAS). ---- */
int mi3pt_read_texture(mi3pt_ctx *ctx, int which /* mi3pt_texture */, float *dst, size_t nfloats);
int mi3pt_read_canvas_rgba8(mi3pt_ctx *ctx, uint8_t *dst, size_t nbytes);
/* Host -> device: load a running mean into the accumulation image (which = ACCUMULATION or
 * OUTPUT; both name the same image after an accumulate pass, accumulate.ts:171-175).  For
 * checkpoint / resume and for presenting a gathered multi-GPU image from a 1-rank context. */
int mi3pt_write_texture(mi3pt_ctx *ctx, int which /* mi3pt_texture */, const float *src, size_t nfloats);

/* Device pointer of the accumulation image (local_rows x width x 4 fp32), for
 * zero-copy hand-off to a collective (RCCL gather of the HDR buffer). */
int mi3pt_accumulation_device_ptr(mi3pt_ctx *ctx, void **dev_ptr, size_t *nbytes);
/* Use caller-owned device memory (e.g. a torch tensor) as the accumulation image;
 * nbytes must equal local_rows*width*16.  NULL returns to the internal buffer. */
int mi3pt_bind_accumulation(mi3pt_ctx *ctx, void *dev_ptr, size_t nbytes);

/* ---- timing: TimingHelper / RollingAverage, timing.ts:28-146, pass.ts:22-26.
 * GPU time of the pass in the most recent completed submit, microseconds
 * (hipEvent pair on the context's stream).  Off by default; enabling costs one
 * event pair per pass per submit. ---- */
int mi3pt_enable_timing(mi3pt_ctx *ctx, int enabled);
int mi3pt_pass_time_us(mi3pt_ctx *ctx, int pass, float *microseconds);
/* With timing enabled: GPU time summed over the batched raytrace launches since the last
 * reset (HIP event pairs on the stream each kernel ran on), their number, and the frames
 * they covered.  Waits for the launches in flight. */
int mi3pt_raytrace_launch_stats(mi3pt_ctx *ctx, int reset, double *total_ms, uint64_t *launches, uint64_t *frames);
/* GPU-clock span from the start of the first to the end of the last of those launches.  Launches
 * overlap at their tails, so span / launches (not total_ms / launches) is what a launch costs. */
int mi3pt_raytrace_launch_span(mi3pt_ctx *ctx, double *span_ms);

/* ---- counters (roofline inputs) ---- */
int mi3pt_get_counters(mi3pt_ctx *ctx, uint64_t out[MI3PT_CNT_COUNT]);
int mi3pt_reset_counters(mi3pt_ctx *ctx);

/* Kernel variant (0 .. 14): 0 = auto, 1 = per-pixel kernel walking the uploaded records,
 * 2 = per-pixel kernel walking node packets, 3 = persistent waves with lane refill,
 * 4 = persistent per-lane state machine (the default), 5 = 4 with a walk threshold of 48,
 * 6 = 4 with the top 32 node packets staged in LDS (measured: no gain),
 * 7 = 4 with leaf tests deferred into triangle steps of their own; needs a proper tree whose
 * worst-case stack stays below 29 entries (checked at upload), otherwise 4 runs; 8 = 7 with the
 * LDS-staged top of the tree of 6 (measured: no gain); 9 = 7 with exact-image distance culling: a
 * child box that starts farther away than the closest hit so far, by a margin PROVEN to cover the
 * rounding of the reference's fp32 Moller-Trumbore code, is skipped, children are
 * visited near first -- the image is bit-identical, the box / triangle-test COUNTERS are lower than
 * the reference walk's (which has no such bound, raytrace.wgsl:118-203); 10 = 9 on 4-ary "wide" packets
 * (internal children absorbed into their parent where boxes nest: the same leaves are reached -- the
 * fp32 slab test is monotone under nesting -- in half the node steps).  9 and 10 need a proper tree whose
 * order-independent worst-case stack fits 56 entries.
 * 11 = 10 with the FILTERED slab test in the shipped batched launch: each box is decided from approximate quotients
 * (one multiplication instead of an exact division each) whenever the two ends of the approximate interval lie more
 * than 2^-21 (relative) apart, which PROVES the reference's decision (csrc/pt_kernels.hip slab_q0; tests/test_slab_filter.py);
 * any other box runs the exact test.  12 = 11 with the culling condition evaluated on one axis only (cheaper, skips
 * less).  11 and 12 exist in the experiment build only (superseded by 13; a release library runs 10 for them).
 * 13 = THE SHIPPED WALK (round 4): 10 on COMPRESSED wide packets -- 64 bytes per 4-ary node: a grid origin, three cell exponents
 * and the four child boxes as 8-bit cell indices rounded OUTWARD by at least one cell -- and 64-byte triangle records that carry
 * the leaf's own box as uploaded.  Above the leaves a box test only has to never reject what the reference's exact test passes
 * (boxes nest: a leaf that passes has every ancestor pass), so the node step runs a conservative test on the decoded planes
 * (one v_cvt_f32_ubyte + one fma per quotient); the leaf's own box is tested EXACTLY in the triangle step, and only leaves that
 * pass it count as triangle tests.  Half the bytes and half the loads of 10 per node step.  Instantiations: culling condition
 * (one axis / three axes, chosen per scene like 12 / 11) x walk threshold (32 lanes; 44 for very large trees) x waves per SIMD (six:
 * 80 registers, for launches of >= 1.5 M jobs; five: 96 registers; MI3PT_OPT_SIX_WAVES).
 * Needs every box of the tree nested in its parent's and finite (any tree of the reference's builder); otherwise 10 runs.
 * 14 = the EIGHT-wide walk (round 6; measured, NOT the default -- profiles/r06_a_ab_eight_wide.log): 13's conservative test on 80-byte packets
 * of up to eight children whose slots are chosen by position, so that `slot ^ octant of the ray's direction signs` is a front-to-back
 * order: a node step leaves two 8-bit hit masks, pushes at most ONE 64-bit node entry (first child packet, hits in visiting order,
 * internal-slot mask) and ONE 32-bit leaf entry (record base, hit slots) and pops the nearest child by a find-first-bit -- no sort,
 * no per-child references, a stack of one entry per tree level.  26 % fewer dependent node steps per ray, 47 % more box tests,
 * 1.4 x the instructions per step: -2 .. -5 % on the 870 k-triangle scene, -25 % on the 10 M-triangle forest.  Same bits (the
 * grouping of the reference tree's nodes into packets is free: only the leaves' own boxes decide what is tested).  Falls back to 13
 * where its packets could not be built (a tree more than 30 packet levels deep, more than 2^24 packets or records).
 * Every reference-legal setting (samplesPerFrame 1 .. 2^24, maxBounces 0 .. 65535, F16 storage, pipelining off) runs this
 * kernel; launches beyond those packing limits run the per-pixel kernel (2), frame by frame.
 * auto = 13 when the scene allows the wide walk and its compressed packets could be built -- else 10 (an experiment build: 10,
 * 11 or 12 by the tree's statistics), else 9, else 7, else 4; MI3PT_OPT_WIDE / MI3PT_OPT_CULL = 0 make auto stop at 9 / 7.
 * Variants 1-8 execute exactly the reference's tests (counters equal the oracle's); all variants produce the same bits.
 * mi3pt_debug_active_variant says what a submit would run now, mi3pt_debug_last_launch what the last one ran. */
int mi3pt_set_kernel_variant(mi3pt_ctx *ctx, int variant);
/* Scheduling options: how the same work is cut into launches, steps and jobs.  NONE of them changes a bit of any image
 * or a path counter (tests/test_gpu_parity.py holds each to the defaults' output); they exist for the sweeps under
 * profiles/ and for the tests.  The library reads NO environment variable for them: a build with -DMI3PT_EXPERIMENTS
 * (make -C webgpu-pathtracer_amd/csrc experiments -> libmi3pt_exp.so) maps MI3PT_<NAME> variables onto this call at
 * mi3pt_create and adds two switches that DO change what is computed (plain-division slab tests; a scaled culling margin,
 * which voids its proof) -- those exist in no release object. */
typedef enum mi3pt_option {
    MI3PT_OPT_WALK_MIN = 0,    /* walk while at least this many lanes are walking (32) */
    MI3PT_OPT_LEAF_MIN = 1,    /* run a triangle step once this many lanes have a leaf parked (24) */
    MI3PT_OPT_SHADE_SPLIT = 2, /* service step: serve the larger of the hit / miss groups, the other only with >= n lanes (64) */
    MI3PT_OPT_TAIL_POLICY = 3, /* drain-phase scheduling bits (7) */
    MI3PT_OPT_TOP_PACKETS = 4, /* kernel variants 6 / 8: node packets staged in LDS per wave */
    MI3PT_OPT_TRI_PAIR = 5,    /* a triangle step tests two parked triangles of a lane that has two (1) */
    MI3PT_OPT_JOB_REVERSE = 6, /* the launch's jobs bottom band first (1) */
    MI3PT_OPT_JOB_GROUP = 7,   /* tiles per job group; 0 frame-major, -1 the library's choice (-1) */
    MI3PT_OPT_JOB_CHUNK = 8,   /* job tickets per draw while the queue is long (4) */
    MI3PT_OPT_BATCH_LIMIT = 9, /* upper bound of frames per launch, of this context -- or of every member of this group (512) */
    MI3PT_OPT_BATCH = 10,      /* frames per launch on one GPU; x nranks for a rank of a tile split (256; the product is capped by MI3PT_OPT_BATCH_LIMIT); 1 = no batching */
    MI3PT_OPT_WAVES_PER_CU = 11, /* resident one-wave workgroups per compute unit (0 = what the kernel build is compiled for: 24 / 20 for the shipped walk's six- / five-wave builds, 16 otherwise) */
    MI3PT_OPT_CULL = 12,       /* 0: `auto` stops at variant 7 */
    MI3PT_OPT_WIDE = 13,       /* 0: `auto` stops at variant 9 */
    MI3PT_OPT_GATE = 14,       /* launches wait for their predecessor's drain mark (1; 0 when a profiler is attached) */
    MI3PT_OPT_GATE_TIMEOUT_MS = 22, /* the wait above has no bound of its own: a blocking entry point (mi3pt_sync, reads, pass times) that has
                                   * waited this long with a launch still held publishes the mark from the host (2000; 0 = never).  An early
                                   * release only lets two launches overlap more -- same bits.  The gate switches itself off, with a
                                   * "warning: ..." text in mi3pt_last_error() and MI3PT_OK returned, once a predecessor is seen to have
                                   * finished without publishing or after the third release */
    MI3PT_OPT_GATE_RELEASES = 23,   /* READ-ONLY: host-side releases so far */
    MI3PT_OPT_CAMERA_BASE = 25,     /* batched launches load the pixel-only part of a camera ray (uv, cameraToRay's direction, cam_pos + dir0 x
                                   * focalDistance) from an image formed once per camera, size and tile instead of forming it in every frame of
                                   * every pixel (1); costs 2 x 16 bytes per pixel of this context's share; 0 = formed in the kernel */
    MI3PT_OPT_PACKET_ORDER = 26,    /* numbering of the 4-ary packets in device memory: 0 breadth-first (the reference's flattenBVH order carried
                                   * over), 1 depth-first, 2 treelets of three levels; the walk follows references, any numbering renders the
                                   * same bits.  Applied at the next scene analysis (0) */
    MI3PT_OPT_SIX_WAVES = 27,       /* the shipped walk's build: 1 = six waves per SIMD (80 registers, a 19-entry LDS stack -- 25 for very large trees,
                                   * whose parked path state then lives in memory), 0 = five (96 registers, 24 entries), -1 = by the size of
                                   * the launch: six from 1.5 M jobs (tiles x frames) on -- long launches gain 2 .. 5 % from the extra wave, a
                                   * rank of an 8-way split's 10 ms launches lose 2 .. 3 % to the longer drain (-1) */
    MI3PT_OPT_LAST_BUILD = 29,      /* READ-ONLY: which instantiation of the state-machine kernel the most recent raytrace launch ran, beside
                                   * mi3pt_debug_last_launch: waves per SIMD it is compiled for (bits 0-7), the one-axis culling condition (bit 8),
                                   * the walk threshold (bits 16-23) -- the template arguments rocprofv3 prints */
    MI3PT_OPT_WALK_ADAPT = 30,      /* the walk threshold by the VIEW (round 6): a launch of the shipped walk reports what it cost -- box tests per ray --
                                   * through host-visible memory, and later launches run the deep-walk build (made for very large trees: walk_min 44,
                                   * a longer leaf list) while that is above 40, the ordinary one again below 30: the 870 k-triangle scene from close
                                   * up +4.4 %, its stated view (17 boxes per ray) unchanged.  No wait, no effect on any bit; MI3PT_OPT_WALK_MIN != 0
                                   * overrides it (1) */
    MI3PT_OPT_COLLAPSE = 28,        /* how the reference tree's nodes are grouped into the walks' wide packets: 1 = the SAH-optimal collapse (round 6:
                                   * fewest expected packet visits; 5.9 instead of 4.0 children per 8-ary packet), 0 = rounds 2 - 5's greedy one (open
                                   * the child with the largest area until the packet is full: packets of two at the bottom of the tree), -1 = greedy
                                   * for the 4-ary packets, optimal for the 8-ary ones: measured, the fuller packets change the boxes a ray tests by
                                   * < 1 % -- the half-empty packets are the ones rays seldom reach (profiles/r06_g_collapse.log).  Same bits (-1) */
    MI3PT_OPT_DEBUG_SUPPRESS_DRAIN = 24, /* tests: arm the gate but let no kernel publish its mark (forces the situation the time-out exists for) */
    MI3PT_OPT_SLOT_SETS = 15,  /* sets of per-frame radiance slots, 2 or 3; before mi3pt_resize (2) */
    MI3PT_OPT_PIPELINE = 16,   /* = mi3pt_set_pipelining */
    MI3PT_OPT_PRESENT_DEPTH = 18, /* MI3PT_PRESENT_EXACT: presenting frames that share one raytrace launch, >= 1 (16); each still
                                   * gets its own accumulate and fullscreen pass, in order -- 1 = also its own launch */
    MI3PT_OPT_HOST_ANALYSES = 19, /* READ-ONLY: host-side scene compiles this context has done (uploads that build packets, relabellings, cull
                                   * analyses); for a device group the sum over its members -- one per scene change whatever the group's
                                   * size: the scene is compiled by member 0 and copied device to device (tests/test_gpu_group.py) */
    MI3PT_OPT_DIAG_LITE = 21,     /* experiment build: with mi3pt_debug_wave_times enabled, run the LEAN kernel with lane counts per kind of step
                                   * (five waves per SIMD, like the shipped one) instead of the four-wave diagnostic twin (profiles/wave_timeline.py) */
    MI3PT_OPT_GATHER_STAGED = 20, /* device group: 1 = every member's rows reach the presenting context through pinned host memory -- the
                                   * path the gather takes where peer access is unavailable or a direct copy failed (forced: tests) */
    MI3PT_OPT_COST_ORDER = 17  /* a launch's jobs in the order of the tiles' measured cost, costliest first: one launch adds up the path
                                * segments per 8x8 tile, later launches with the same uniforms run the
                                * cheapest quarter of the tiles last (1), or every tile costliest first (2).  Default 0: measured +0.3 % on
                                * one GPU and -1.6 ... -4 % for a rank of a tile split in batched launches; 2 is for hosts that launch
                                * and wait frame by frame: 1.29 -> 1.205 ms per 1080p frame of the 870 k-triangle scene, the demo
                                * scene +-0 (profiles/r04_r_interactive_cost_order.log) */
} mi3pt_option;
int mi3pt_debug_set_option(mi3pt_ctx *ctx, int option /* mi3pt_option */, int value);
int mi3pt_debug_get_option(mi3pt_ctx *ctx, int option /* mi3pt_option */, int *value);
/* The variant a raytrace submit would run right now: the selected one, or its fall-back when the
 * uploaded scene does not admit it (tests use it to make sure nothing fell back silently). */
int mi3pt_debug_active_variant(mi3pt_ctx *ctx, int *variant);
/* The kernel the most recent raytrace launch actually ran: kind 0 = per-pixel kernel (one 8x8 tile per wave, the WGSL control
 * flow: variants 1 / 2, and the fall-back for launches beyond the state-machine kernel's packing limits), 1 = the persistent
 * state-machine kernel; its variant after the scene / launch fall-backs; lean = 1: the build without diagnostics that every
 * ordinary launch runs (96 vector registers, five waves per SIMD) -- whatever samplesPerFrame, maxBounces, storage format,
 * pipelining and presentation mode are (tests/test_gpu_parity.py::test_every_reference_setting_runs_the_lean_kernel); 0: the
 * diagnostic twin (a diagnostic buffer is bound or a step-voting option was changed through mi3pt_debug_set_option);
 * workgroups = the launch's grid.  Any pointer may be NULL. */
int mi3pt_debug_last_launch(mi3pt_ctx *ctx, int *kind, int *variant, int *lean, int *workgroups);
/* The reference's environment importance sampling (raytrace.wgsl:315-367: getEnvironmentMapUV /
 * ...MarginalCDF / ...ConditionalCDF / ...PDF over the CDF texture of renderer.ts:159-266) is dead
 * code as shipped -- its call sites raytrace.wgsl:398 and :402-404 are commented out.  enabled = 1
 * runs the frame as if those three lines were live (a miss takes its uv from the CDF and divides
 * the path's light by the pdf); it needs the CDF texture and uses the per-pixel kernel (no
 * batching).  Default 0 = the shipped behaviour. */
int mi3pt_set_env_sampling(mi3pt_ctx *ctx, int enabled);
/* Frame pipelining (default on): RAYTRACE|ACCUMULATE submits are queued; up to 256 consecutive
 * frames (mi3pt_batch_capacity) whose uniforms differ only in `frame` run as one raytrace launch plus one ordered
 * multi-frame running mean, and launches alternate between two internal streams so the next
 * one fills the CUs while the last paths of this one drain.  Any call that observes or
 * changes device state flushes the queue first; results are bit-identical.  Off: one fused
 * kernel per frame on the context's stream, launched inside mi3pt_submit. */
int mi3pt_set_pipelining(mi3pt_ctx *ctx, int enabled);

/* ---- component probes on the device (parity tests of the pieces) ----
 * rays: n x 6 floats (origin, direction); out: n x 12 floats
 * (hit, t, position xyz, normal xyz, materialIndex, box tests, triangle tests,
 * stack overflows) = raySceneIntersect, raytrace.wgsl:205-211. */
int mi3pt_debug_intersect(mi3pt_ctx *ctx, const float *rays, size_t n, float *out);
/* An ALTERNATIVE tree, built on the device from the uploaded triangles: a linear
 * BVH (Morton codes, radix sort, Karras' parallel hierarchy, bottom-up fit) in the same 48-byte
 * records, pre-order numbered so that a child follows its parent; nodes_out (host) receives
 * (2*ntris - 1) x 48 B for mi3pt_upload_bvh.  It is NOT the reference's SAH tree
 * (mi3pt_host_build_bvh_f64 is): box / triangle test counts differ, the image does not, except
 * where two triangles are hit at exactly the same t.  build_ms (optional): device time. */
int mi3pt_device_build_bvh(mi3pt_ctx *ctx, void *nodes_out, size_t nodes_capacity_bytes, size_t *nnodes_out, float *build_ms);

/* Debug: node-packet / triangle numbering used on the device.  0 (default, shipped) = packets
 * breadth-first like the uploaded nodes, triangles as uploaded.  1 = packets in the reference's
 * visiting order (node, right subtree, left subtree) and triangles in leaf-visiting order -- a
 * pure relabelling, bit-identical by construction, kept as a test of exactly that.  Call before
 * the uploads it should apply to; after switching back, the next upload of the BVH or of the
 * triangles restores the shipped numbering. */
int mi3pt_debug_set_packet_layout(mi3pt_ctx *ctx, int layout);

/* fn: 0 sin, 1 cos, 2 tan, 3 log, 4 exp, 5 atan2(a,b), 6 asin, 7 pow(a,b),
 * 8 fp16 round trip, 9 sqrt, 10 a/b; the kernels' reduced-instruction forms: 11 sqrt, 12 1/a,
 * 13..15 x/y/z of normalize(a, b, a - b), 16 log for a = 0 or a in [2^-32, 1] (rand()'s values: what randNormal
 * calls).  b may be NULL for unary functions. */
int mi3pt_debug_math(mi3pt_ctx *ctx, int fn, const float *a, const float *b, float *out, size_t n);

/* Diagnostic record of the last persistent raytrace launch: 16 x uint64 per resident wave
 * (begin, work-queue-empty, end on the 100 MHz wall clock; shader cycles begin->end; then
 * packed step statistics: walk steps | walking lanes, service steps | leaf lanes,
 * shaded lanes | hit lanes, path starts | segment starts; [8] triangle steps of the
 * deferred-leaf walk; [9..11] shader cycles spent in node / triangle / service steps; [12..15] reserved).
 * out == NULL: enable != 0 allocates the buffer, enable == 0 frees it. */
int mi3pt_debug_wave_times(mi3pt_ctx *ctx, int enable, uint64_t *out, size_t capacity_slots, size_t *slots_out);

/* ---- host-side scene compile (CPU, no device needed) ----
 * mi3pt_host_build_bvh: buildBVH + buildBVHRecursive + flattenBVH,
 * raytrace.ts:540-694, producing the identical tree (same axis rule, stable sort,
 * first strict SAH minimum, breadth-first flatten).  `triangles` is ntris x 112 B;
 * `nodes_out` receives (2*ntris - 1) x 48 B.  Returns the node count through
 * *nnodes_out.  nthreads <= 0 picks the hardware concurrency. */
int mi3pt_host_build_bvh(const void *triangles, size_t ntris, void *nodes_out,
                         size_t nodes_capacity_bytes, size_t *nnodes_out, int nthreads);
/* Same builder on the JavaScript host's double-precision world-space positions
 * (9 doubles per triangle: a.xyz b.xyz c.xyz), i.e. before the structured view
 * rounds them to fp32 -- this is what raytrace.ts:546-550 feeds Box3.setFromPoints,
 * so near-tie splits come out exactly as in the reference. */
int mi3pt_host_build_bvh_f64(const double *positions, size_t ntris, void *nodes_out,
                             size_t nodes_capacity_bytes, size_t *nnodes_out, int nthreads);
/* updateEnvironmentTexture's CDF texture, renderer.ts:159-266: R marginal CDF,
 * G conditional CDF, B sin-weighted luminance, A 1. */
int mi3pt_host_env_cdf(const float *rgba, int width, int height, float *cdf_rgba_out);
/* Host-only self-check of the eight-wide packets of kernel variant 14 (no device; what the CPU tests call): builds them for a tree
 * (48-byte records) + triangles (112-byte records) as the scene analysis does (greedy = 0: the SAH-optimal grouping, what a context uses by
 * default for these packets; 1: the greedy one, MI3PT_OPT_COLLAPSE = 0), then walks the result independently of the builder --
 * every leaf triangle reachable exactly once, every packet referenced once, records carrying their leaf's box and triangle, every
 * decoded child box containing everything below it.  out[0..5] = packets, records, packet levels, children per packet x 1000,
 * leaves reached, 1 if a context would offer variant 14 for this tree (levels within the walk's stack).  MI3PT_ERR_STATE with the
 * reason in mi3pt_last_error when the tree does not admit the packets or a check fails. */
int mi3pt_host_eight_wide_check(const void *nodes, size_t nodes_bytes, const void *triangles, size_t triangles_bytes, int greedy, uint64_t out[6]);

#ifdef __cplusplus
}
#endif
#endif /* MI3PT_H */
