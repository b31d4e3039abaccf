// mi3pt_napi.cc -- thin N-API shim over the C ABI of libmi3pt.so (include/mi3pt.h).
//
// One JS function per C entry point; typed arrays / Buffers in and out; a non-zero
// status becomes a thrown Error carrying mi3pt_last_error() (the reference throws at
// the same places: src/renderer.ts:65-67, 133-143, 514-516; src/passes/raytrace.ts:563-565).
// No computation happens here.  Built with plain g++ against /usr/include/node (no
// node-gyp, no network): see Makefile.
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../../include/mi3pt.h"

namespace {

#define NAPI_OK(call)                                                     \
    do {                                                                  \
        if ((call) != napi_ok) {                                          \
            napi_throw_error(env, nullptr, "N-API call failed: " #call);  \
            return nullptr;                                               \
        }                                                                 \
    } while (0)

napi_value throw_status(napi_env env, int rc)
{
    char code[16];
    std::snprintf(code, sizeof code, "MI3PT_%d", rc);
    napi_throw_error(env, code, mi3pt_last_error());
    return nullptr;
}

#define MI3PT_TRY(expr)                            \
    do {                                           \
        int rc_ = (expr);                          \
        if (rc_ != MI3PT_OK) return throw_status(env, rc_); \
    } while (0)

struct Args {
    napi_value v[8];
    size_t n = 8;
};

bool get_args(napi_env env, napi_callback_info info, Args &a, size_t need)
{
    a.n = 8;
    if (napi_get_cb_info(env, info, &a.n, a.v, nullptr, nullptr) != napi_ok || a.n < need) {
        napi_throw_type_error(env, nullptr, "too few arguments");
        return false;
    }
    return true;
}

bool get_ctx(napi_env env, napi_value v, mi3pt_ctx **out)
{
    void *p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, nullptr, "expected a context handle");
        return false;
    }
    mi3pt_ctx **slot = static_cast<mi3pt_ctx **>(p);
    if (!*slot) {
        napi_throw_error(env, nullptr, "context has been destroyed");
        return false;
    }
    *out = *slot;
    return true;
}

bool get_i32(napi_env env, napi_value v, int32_t *out)
{
    if (napi_get_value_int32(env, v, out) != napi_ok) {
        napi_throw_type_error(env, nullptr, "expected a number");
        return false;
    }
    return true;
}

// Bytes of a Buffer / TypedArray / ArrayBuffer / DataView.
bool get_bytes(napi_env env, napi_value v, void **data, size_t *nbytes)
{
    bool is = false;
    if (napi_is_buffer(env, v, &is) == napi_ok && is) return napi_get_buffer_info(env, v, data, nbytes) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type t;
        size_t len;
        napi_value ab;
        size_t off;
        if (napi_get_typedarray_info(env, v, &t, &len, data, &ab, &off) != napi_ok) return false;
        static const size_t width[] = { 1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8 };
        *nbytes = len * width[t];
        return true;
    }
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, nbytes) == napi_ok;
    if (napi_is_dataview(env, v, &is) == napi_ok && is) {
        napi_value ab;
        size_t off;
        return napi_get_dataview_info(env, v, nbytes, data, &ab, &off) == napi_ok;
    }
    napi_throw_type_error(env, nullptr, "expected a Buffer, TypedArray or ArrayBuffer");
    return false;
}

napi_value undefined(napi_env env)
{
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

void finalize_ctx(napi_env, void *data, void *)
{
    mi3pt_ctx **slot = static_cast<mi3pt_ctx **>(data);
    if (*slot) mi3pt_destroy(*slot);
    delete slot;
}

// ---- library / devices

napi_value AbiVersion(napi_env env, napi_callback_info)
{
    napi_value r;
    NAPI_OK(napi_create_int32(env, mi3pt_abi_version(), &r));
    return r;
}

napi_value DeviceCount(napi_env env, napi_callback_info)
{
    int n = 0;
    MI3PT_TRY(mi3pt_device_count(&n));
    napi_value r;
    NAPI_OK(napi_create_int32(env, n, &r));
    return r;
}

napi_value DeviceName(napi_env env, napi_callback_info info)
{
    Args a;
    int32_t dev;
    if (!get_args(env, info, a, 1) || !get_i32(env, a.v[0], &dev)) return nullptr;
    char name[256];
    MI3PT_TRY(mi3pt_device_name(dev, name, sizeof name));
    napi_value r;
    NAPI_OK(napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &r));
    return r;
}

napi_value TileLocalRows(napi_env env, napi_callback_info info)
{
    Args a;
    int32_t h, rank, nranks, block;
    if (!get_args(env, info, a, 4) || !get_i32(env, a.v[0], &h) || !get_i32(env, a.v[1], &rank) ||
        !get_i32(env, a.v[2], &nranks) || !get_i32(env, a.v[3], &block))
        return nullptr;
    napi_value r;
    NAPI_OK(napi_create_int32(env, mi3pt_tile_local_rows(h, rank, nranks, block), &r));
    return r;
}

napi_value TileGlobalRow(napi_env env, napi_callback_info info)
{
    Args a;
    int32_t ly, rank, nranks, block;
    if (!get_args(env, info, a, 4) || !get_i32(env, a.v[0], &ly) || !get_i32(env, a.v[1], &rank) ||
        !get_i32(env, a.v[2], &nranks) || !get_i32(env, a.v[3], &block))
        return nullptr;
    napi_value r;
    NAPI_OK(napi_create_int32(env, mi3pt_tile_global_row(ly, rank, nranks, block), &r));
    return r;
}

napi_value TileOwner(napi_env env, napi_callback_info info)
{
    Args a;
    int32_t y, nranks, block;
    if (!get_args(env, info, a, 3) || !get_i32(env, a.v[0], &y) || !get_i32(env, a.v[1], &nranks) || !get_i32(env, a.v[2], &block))
        return nullptr;
    napi_value r;
    NAPI_OK(napi_create_int32(env, mi3pt_tile_owner(y, nranks, block), &r));
    return r;
}

// ---- context

napi_value Create(napi_env env, napi_callback_info info)
{
    Args a;
    int32_t dev = 0;
    if (!get_args(env, info, a, 0)) return nullptr;
    if (a.n >= 1 && !get_i32(env, a.v[0], &dev)) return nullptr;
    mi3pt_ctx *ctx = nullptr;
    MI3PT_TRY(mi3pt_create(dev, &ctx));
    mi3pt_ctx **slot = new mi3pt_ctx *(ctx);
    napi_value ext;
    if (napi_create_external(env, slot, finalize_ctx, nullptr, &ext) != napi_ok) {
        mi3pt_destroy(ctx);
        delete slot;
        napi_throw_error(env, nullptr, "napi_create_external failed");
        return nullptr;
    }
    return ext;
}

// createGroup(devices: number[], blockRows = 8): Renderer.create({ devices }) -- mi3pt_create_group; the handle goes through
// the same functions as a single-device one
napi_value CreateGroup(napi_env env, napi_callback_info info)
{
    Args a;
    if (!get_args(env, info, a, 1)) return nullptr;
    bool is_array = false;
    uint32_t n = 0;
    if (napi_is_array(env, a.v[0], &is_array) != napi_ok || !is_array || napi_get_array_length(env, a.v[0], &n) != napi_ok || n < 1 || n > 64) {
        napi_throw_type_error(env, nullptr, "createGroup: an array of 1..64 device indices");
        return nullptr;
    }
    int devices[64];
    for (uint32_t i = 0; i < n; i++) {
        napi_value e;
        int32_t d = 0;
        if (napi_get_element(env, a.v[0], i, &e) != napi_ok || !get_i32(env, e, &d)) return nullptr;
        devices[i] = d;
    }
    int32_t block_rows = 8;
    if (a.n >= 2 && !get_i32(env, a.v[1], &block_rows)) return nullptr;
    mi3pt_ctx *ctx = nullptr;
    MI3PT_TRY(mi3pt_create_group(devices, (int)n, block_rows, &ctx));
    mi3pt_ctx **slot = new mi3pt_ctx *(ctx);
    napi_value ext;
    if (napi_create_external(env, slot, finalize_ctx, nullptr, &ext) != napi_ok) {
        mi3pt_destroy(ctx);
        delete slot;
        napi_throw_error(env, nullptr, "napi_create_external failed");
        return nullptr;
    }
    return ext;
}

napi_value Destroy(napi_env env, napi_callback_info info)
{
    Args a;
    if (!get_args(env, info, a, 1)) return nullptr;
    void *p = nullptr;
    if (napi_get_value_external(env, a.v[0], &p) != napi_ok || !p) return undefined(env);
    mi3pt_ctx **slot = static_cast<mi3pt_ctx **>(p);
    if (*slot) {
        mi3pt_destroy(*slot);
        *slot = nullptr;
    }
    return undefined(env);
}

#define CTX_INT_FN(NAME, CALL)                                                             \
    napi_value NAME(napi_env env, napi_callback_info info)                                \
    {                                                                                      \
        Args a;                                                                            \
        mi3pt_ctx *ctx;                                                                    \
        int32_t x;                                                                         \
        if (!get_args(env, info, a, 2) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &x)) return nullptr; \
        MI3PT_TRY(CALL(ctx, x));                                                           \
        return undefined(env);                                                             \
    }
#define CTX_VOID_FN(NAME, CALL)                                                            \
    napi_value NAME(napi_env env, napi_callback_info info)                                \
    {                                                                                      \
        Args a;                                                                            \
        mi3pt_ctx *ctx;                                                                    \
        if (!get_args(env, info, a, 1) || !get_ctx(env, a.v[0], &ctx)) return nullptr;     \
        MI3PT_TRY(CALL(ctx));                                                              \
        return undefined(env);                                                             \
    }
#define CTX_BYTES_FN(NAME, CALL)                                                           \
    napi_value NAME(napi_env env, napi_callback_info info)                                \
    {                                                                                      \
        Args a;                                                                            \
        mi3pt_ctx *ctx;                                                                    \
        void *data;                                                                        \
        size_t n;                                                                          \
        if (!get_args(env, info, a, 2) || !get_ctx(env, a.v[0], &ctx) || !get_bytes(env, a.v[1], &data, &n)) return nullptr; \
        MI3PT_TRY(CALL(ctx, data, n));                                                     \
        return undefined(env);                                                             \
    }

CTX_INT_FN(SetStorage, mi3pt_set_storage)
CTX_INT_FN(SetKernelVariant, mi3pt_set_kernel_variant)
CTX_INT_FN(EnableTiming, mi3pt_enable_timing)
CTX_INT_FN(SetPipelining, mi3pt_set_pipelining)
CTX_INT_FN(SetPresentMode, mi3pt_set_present_mode)
CTX_INT_FN(SetEnvSampling, mi3pt_set_env_sampling)
CTX_VOID_FN(Reset, mi3pt_reset)
CTX_VOID_FN(Sync, mi3pt_sync)
CTX_VOID_FN(Flush, mi3pt_flush)
CTX_VOID_FN(ResetCounters, mi3pt_reset_counters)
CTX_BYTES_FN(UploadTriangles, mi3pt_upload_triangles)
CTX_BYTES_FN(UploadMaterials, mi3pt_upload_materials)
CTX_BYTES_FN(UploadBvh, mi3pt_upload_bvh)

napi_value SetTile(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t rank, nranks, block;
    if (!get_args(env, info, a, 4) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &rank) ||
        !get_i32(env, a.v[2], &nranks) || !get_i32(env, a.v[3], &block))
        return nullptr;
    MI3PT_TRY(mi3pt_set_tile(ctx, rank, nranks, block));
    return undefined(env);
}

napi_value upload_env(napi_env env, napi_callback_info info, bool cdf)
{
    Args a;
    mi3pt_ctx *ctx;
    void *data;
    size_t n;
    int32_t w, h;
    if (!get_args(env, info, a, 4) || !get_ctx(env, a.v[0], &ctx) || !get_bytes(env, a.v[1], &data, &n) ||
        !get_i32(env, a.v[2], &w) || !get_i32(env, a.v[3], &h))
        return nullptr;
    if (w > 0 && h > 0 && n != (size_t)w * (size_t)h * 16) {
        napi_throw_range_error(env, nullptr, "environment data must be width*height*4 floats");
        return nullptr;
    }
    MI3PT_TRY(cdf ? mi3pt_upload_environment_cdf(ctx, static_cast<const float *>(data), w, h)
                  : mi3pt_upload_environment(ctx, static_cast<const float *>(data), w, h));
    return undefined(env);
}
napi_value UploadEnvironment(napi_env env, napi_callback_info info) { return upload_env(env, info, false); }
napi_value UploadEnvironmentCdf(napi_env env, napi_callback_info info) { return upload_env(env, info, true); }

napi_value Resize(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t w, h;
    if (!get_args(env, info, a, 3) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &w) || !get_i32(env, a.v[2], &h))
        return nullptr;
    MI3PT_TRY(mi3pt_resize(ctx, w, h));
    return undefined(env);
}

napi_value SetUniforms(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t pass;
    void *data;
    size_t n;
    if (!get_args(env, info, a, 3) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &pass) ||
        !get_bytes(env, a.v[2], &data, &n))
        return nullptr;
    MI3PT_TRY(mi3pt_set_uniforms(ctx, pass, data, n));
    return undefined(env);
}

napi_value Submit(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t mask;
    if (!get_args(env, info, a, 2) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &mask)) return nullptr;
    MI3PT_TRY(mi3pt_submit(ctx, (unsigned)mask));
    return undefined(env);
}

// readTexture(ctx, which, nfloats) -> Float32Array
napi_value ReadTexture(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t which, nfloats;
    if (!get_args(env, info, a, 3) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &which) ||
        !get_i32(env, a.v[2], &nfloats) || nfloats < 0)
        return nullptr;
    void *data = nullptr;
    napi_value ab, ta;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)nfloats * 4, &data, &ab));
    MI3PT_TRY(mi3pt_read_texture(ctx, which, static_cast<float *>(data), (size_t)nfloats));
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, (size_t)nfloats, ab, 0, &ta));
    return ta;
}

// writeTexture(ctx, which, Float32Array) : GPUQueue.writeTexture for the HDR images
napi_value WriteTexture(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t which;
    void *data;
    size_t n;
    if (!get_args(env, info, a, 3) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &which) ||
        !get_bytes(env, a.v[2], &data, &n))
        return nullptr;
    MI3PT_TRY(mi3pt_write_texture(ctx, which, static_cast<const float *>(data), n / 4));
    return undefined(env);
}

// raytraceLaunchStats(ctx, reset) -> { totalMs, launches, frames }
napi_value RaytraceLaunchStats(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t reset;
    if (!get_args(env, info, a, 2) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &reset)) return nullptr;
    double ms = 0.0;
    uint64_t launches = 0, frames = 0;
    MI3PT_TRY(mi3pt_raytrace_launch_stats(ctx, reset, &ms, &launches, &frames));
    napi_value obj, v;
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_create_double(env, ms, &v));
    NAPI_OK(napi_set_named_property(env, obj, "totalMs", v));
    NAPI_OK(napi_create_double(env, (double)launches, &v));
    NAPI_OK(napi_set_named_property(env, obj, "launches", v));
    NAPI_OK(napi_create_double(env, (double)frames, &v));
    NAPI_OK(napi_set_named_property(env, obj, "frames", v));
    return obj;
}

// readCanvasRgba8(ctx, nbytes) -> Uint8Array
napi_value ReadCanvasRgba8(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t nbytes;
    if (!get_args(env, info, a, 2) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &nbytes) || nbytes < 0)
        return nullptr;
    void *data = nullptr;
    napi_value ab, ta;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)nbytes, &data, &ab));
    MI3PT_TRY(mi3pt_read_canvas_rgba8(ctx, static_cast<uint8_t *>(data), (size_t)nbytes));
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, (size_t)nbytes, ab, 0, &ta));
    return ta;
}

napi_value PassTimeUs(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t pass;
    if (!get_args(env, info, a, 2) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &pass)) return nullptr;
    float us = 0.0f;
    int rc = mi3pt_pass_time_us(ctx, pass, &us);
    if (rc == MI3PT_ERR_STATE) {        // pass not timed in the last submit -> null, like NoTimingHelper
        napi_value n;
        napi_get_null(env, &n);
        return n;
    }
    if (rc != MI3PT_OK) return throw_status(env, rc);
    napi_value r;
    NAPI_OK(napi_create_double(env, us, &r));
    return r;
}

napi_value GetCounters(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    if (!get_args(env, info, a, 1) || !get_ctx(env, a.v[0], &ctx)) return nullptr;
    uint64_t c[MI3PT_CNT_COUNT];
    MI3PT_TRY(mi3pt_get_counters(ctx, c));
    static const char *names[] = { "rays", "boxTests", "triTests", "hits", "misses", "stackOverflows", "pixels" };
    napi_value obj;
    NAPI_OK(napi_create_object(env, &obj));
    for (int k = 0; k < 7; k++) {
        napi_value v;
        NAPI_OK(napi_create_double(env, (double)c[k], &v));
        NAPI_OK(napi_set_named_property(env, obj, names[k], v));
    }
    return obj;
}

// ---- host-side scene compile

napi_value make_node_buffer(napi_env env, size_t ntris, void **data)
{
    napi_value buf;
    if (napi_create_buffer(env, (2 * ntris - 1) * MI3PT_BVHNODE_STRIDE, data, &buf) != napi_ok) return nullptr;
    return buf;
}

// hostBuildBvhF64(Float64Array positions [9 per triangle], nthreads) -> Buffer of 48-B nodes
napi_value HostBuildBvhF64(napi_env env, napi_callback_info info)
{
    Args a;
    void *data;
    size_t n;
    int32_t threads = 0;
    if (!get_args(env, info, a, 1) || !get_bytes(env, a.v[0], &data, &n)) return nullptr;
    if (a.n >= 2 && !get_i32(env, a.v[1], &threads)) return nullptr;
    const size_t ntris = n / 72;
    if (ntris == 0 || n % 72) {
        napi_throw_error(env, nullptr, "Input nodes array is empty");       // raytrace.ts:563-565
        return nullptr;
    }
    void *out = nullptr;
    napi_value buf = make_node_buffer(env, ntris, &out);
    if (!buf) return nullptr;
    size_t count = 0;
    MI3PT_TRY(mi3pt_host_build_bvh_f64(static_cast<const double *>(data), ntris, out,
                                       (2 * ntris - 1) * MI3PT_BVHNODE_STRIDE, &count, threads));
    return buf;
}

// deviceBuildBvh(ctx, ntris) -> Buffer of (2*ntris - 1) 48-byte nodes: the linear BVH built on the device
// from the uploaded triangles (an alternative to the reference's SAH tree, see mi3pt.h)
napi_value DeviceBuildBvh(napi_env env, napi_callback_info info)
{
    Args a;
    mi3pt_ctx *ctx;
    int32_t ntris;
    if (!get_args(env, info, a, 2) || !get_ctx(env, a.v[0], &ctx) || !get_i32(env, a.v[1], &ntris) || ntris <= 0) return nullptr;
    void *out = nullptr;
    napi_value buf = make_node_buffer(env, (size_t)ntris, &out);
    if (!buf) return nullptr;
    size_t count = 0;
    MI3PT_TRY(mi3pt_device_build_bvh(ctx, out, (2 * (size_t)ntris - 1) * MI3PT_BVHNODE_STRIDE, &count, nullptr));
    return buf;
}

napi_value HostBuildBvh(napi_env env, napi_callback_info info)
{
    Args a;
    void *data;
    size_t n;
    int32_t threads = 0;
    if (!get_args(env, info, a, 1) || !get_bytes(env, a.v[0], &data, &n)) return nullptr;
    if (a.n >= 2 && !get_i32(env, a.v[1], &threads)) return nullptr;
    const size_t ntris = n / MI3PT_TRIANGLE_STRIDE;
    if (ntris == 0 || n % MI3PT_TRIANGLE_STRIDE) {
        napi_throw_error(env, nullptr, "Input nodes array is empty");
        return nullptr;
    }
    void *out = nullptr;
    napi_value buf = make_node_buffer(env, ntris, &out);
    if (!buf) return nullptr;
    size_t count = 0;
    MI3PT_TRY(mi3pt_host_build_bvh(data, ntris, out, (2 * ntris - 1) * MI3PT_BVHNODE_STRIDE, &count, threads));
    return buf;
}

// hostEnvCdf(Float32Array rgba, width, height) -> Float32Array
napi_value HostEnvCdf(napi_env env, napi_callback_info info)
{
    Args a;
    void *data;
    size_t n;
    int32_t w, h;
    if (!get_args(env, info, a, 3) || !get_bytes(env, a.v[0], &data, &n) || !get_i32(env, a.v[1], &w) ||
        !get_i32(env, a.v[2], &h))
        return nullptr;
    if (w <= 0 || h <= 0 || n != (size_t)w * h * 16) {
        napi_throw_range_error(env, nullptr, "environment data must be width*height*4 floats");
        return nullptr;
    }
    void *out = nullptr;
    napi_value ab, ta;
    NAPI_OK(napi_create_arraybuffer(env, n, &out, &ab));
    MI3PT_TRY(mi3pt_host_env_cdf(static_cast<const float *>(data), w, h, static_cast<float *>(out)));
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, n / 4, ab, 0, &ta));
    return ta;
}

napi_value Init(napi_env env, napi_value exports)
{
    struct { const char *name; napi_callback fn; } fns[] = {
        { "abiVersion", AbiVersion }, { "deviceCount", DeviceCount }, { "deviceName", DeviceName },
        { "tileLocalRows", TileLocalRows }, { "tileGlobalRow", TileGlobalRow }, { "tileOwner", TileOwner }, { "create", Create }, { "createGroup", CreateGroup }, { "destroy", Destroy },
        { "setStorage", SetStorage }, { "setKernelVariant", SetKernelVariant }, { "setTile", SetTile },
        { "uploadTriangles", UploadTriangles }, { "uploadMaterials", UploadMaterials }, { "uploadBvh", UploadBvh },
        { "uploadEnvironment", UploadEnvironment }, { "uploadEnvironmentCdf", UploadEnvironmentCdf },
        { "resize", Resize }, { "reset", Reset }, { "setUniforms", SetUniforms }, { "submit", Submit },
        { "sync", Sync }, { "flush", Flush }, { "readTexture", ReadTexture }, { "readCanvasRgba8", ReadCanvasRgba8 },
        { "enableTiming", EnableTiming }, { "passTimeUs", PassTimeUs }, { "getCounters", GetCounters },
        { "resetCounters", ResetCounters }, { "hostBuildBvhF64", HostBuildBvhF64 }, { "hostBuildBvh", HostBuildBvh },
        { "hostEnvCdf", HostEnvCdf }, { "setPipelining", SetPipelining }, { "setPresentMode", SetPresentMode }, { "setEnvSampling", SetEnvSampling }, { "deviceBuildBvh", DeviceBuildBvh }, { "writeTexture", WriteTexture },
        { "raytraceLaunchStats", RaytraceLaunchStats },
    };
    for (const auto &f : fns) {
        napi_value v;
        if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &v) != napi_ok) return nullptr;
        if (napi_set_named_property(env, exports, f.name, v) != napi_ok) return nullptr;
    }
    return exports;
}

}  // namespace

NAPI_MODULE(mi3pt, Init)
