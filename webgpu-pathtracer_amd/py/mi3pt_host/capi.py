"""ctypes binding of libmi3pt.so (include/mi3pt.h).

Thin by design: one Python method per C entry point, numpy arrays in and out, status
codes turned into `Mi3ptError` (the way the N-API shim turns them into thrown Errors).
No compute happens here and there is no fallback: if the library or a HIP device is
missing the calls raise.
"""
import ctypes
import os

import numpy as np

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (MI3PT_LIBRARY: another build of the library -- the sweeps under profiles/ point it at the experiment build,
# libmi3pt_exp.so, which reads its options from the environment; read here, by the Python host, not by the library)
LIB_PATH = os.environ.get("MI3PT_LIBRARY") or os.path.join(PKG_ROOT, "libmi3pt.so")

PASS_RAYTRACE, PASS_ACCUMULATE, PASS_FULLSCREEN = 0, 1, 2
SUBMIT_RAYTRACE, SUBMIT_ACCUMULATE, SUBMIT_FULLSCREEN = 1, 2, 4
TEX_OUTPUT, TEX_ACCUMULATION, TEX_CANVAS = 0, 1, 2
STORAGE_F32, STORAGE_F16 = 0, 1
PRESENT_EXACT, PRESENT_LATEST = 0, 1
# mi3pt_option (include/mi3pt.h): scheduling options; none changes a bit of any image
(OPT_WALK_MIN, OPT_LEAF_MIN, OPT_SHADE_SPLIT, OPT_TAIL_POLICY, OPT_TOP_PACKETS, OPT_TRI_PAIR, OPT_JOB_REVERSE, OPT_JOB_GROUP,
 OPT_JOB_CHUNK, OPT_BATCH_LIMIT, OPT_BATCH, OPT_WAVES_PER_CU, OPT_CULL, OPT_WIDE, OPT_GATE, OPT_SLOT_SETS, OPT_PIPELINE,
 OPT_COST_ORDER, OPT_PRESENT_DEPTH, OPT_HOST_ANALYSES, OPT_GATHER_STAGED, OPT_DIAG_LITE,
 OPT_GATE_TIMEOUT_MS, OPT_GATE_RELEASES, OPT_DEBUG_SUPPRESS_DRAIN, OPT_CAMERA_BASE, OPT_PACKET_ORDER, OPT_SIX_WAVES, OPT_COLLAPSE, OPT_LAST_BUILD, OPT_WALK_ADAPT) = range(31)
COUNTER_NAMES = ("rays", "box_tests", "tri_tests", "hits", "misses", "stack_overflows", "pixels", "reserved")

# every symbol include/mi3pt.h declares; tests/test_capi_symbols.py checks the header
# against this list and against the built library.
SYMBOLS = (
    "mi3pt_abi_version", "mi3pt_last_error", "mi3pt_device_count", "mi3pt_device_name",
    "mi3pt_create", "mi3pt_destroy", "mi3pt_set_stream", "mi3pt_set_storage", "mi3pt_set_tile",
    "mi3pt_tile_local_rows", "mi3pt_upload_triangles", "mi3pt_upload_materials", "mi3pt_upload_bvh",
    "mi3pt_upload_environment", "mi3pt_upload_environment_cdf", "mi3pt_resize", "mi3pt_reset",
    "mi3pt_set_uniforms", "mi3pt_submit", "mi3pt_sync", "mi3pt_read_texture", "mi3pt_read_canvas_rgba8",
    "mi3pt_write_texture",
    "mi3pt_accumulation_device_ptr", "mi3pt_bind_accumulation", "mi3pt_enable_timing",
    "mi3pt_pass_time_us", "mi3pt_raytrace_launch_stats", "mi3pt_get_counters", "mi3pt_reset_counters", "mi3pt_set_kernel_variant",
    "mi3pt_set_env_sampling", "mi3pt_device_build_bvh",
    "mi3pt_set_pipelining", "mi3pt_flush", "mi3pt_set_present_mode", "mi3pt_raytrace_launch_span", "mi3pt_batch_capacity", "mi3pt_debug_active_variant", "mi3pt_debug_last_launch", "mi3pt_submit_frames", "mi3pt_debug_set_packet_layout",
    "mi3pt_debug_intersect", "mi3pt_debug_math", "mi3pt_debug_wave_times", "mi3pt_host_build_bvh", "mi3pt_host_build_bvh_f64",
    "mi3pt_host_env_cdf", "mi3pt_host_eight_wide_check", "mi3pt_debug_set_option", "mi3pt_debug_get_option",
    "mi3pt_create_group", "mi3pt_group_size", "mi3pt_group_member",
    "mi3pt_tile_global_row", "mi3pt_tile_owner",
)


class Mi3ptError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"mi3pt error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def load_library(path=None):
    """Load libmi3pt.so (built by webgpu-pathtracer_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise Mi3ptError(-1, f"{path} not found: build it with `make -C webgpu-pathtracer_amd/csrc` "
                             "(or __graft_entry__.build())")
    lib = ctypes.CDLL(path)
    c_void_p, c_size_t, c_int = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.mi3pt_last_error.restype = ctypes.c_char_p
    lib.mi3pt_create.argtypes = [c_int, ctypes.POINTER(c_void_p)]
    lib.mi3pt_destroy.argtypes = [c_void_p]
    lib.mi3pt_set_stream.argtypes = [c_void_p, c_void_p]
    lib.mi3pt_set_storage.argtypes = [c_void_p, c_int]
    lib.mi3pt_set_tile.argtypes = [c_void_p, c_int, c_int, c_int]
    lib.mi3pt_set_kernel_variant.argtypes = [c_void_p, c_int]
    lib.mi3pt_set_env_sampling.argtypes = [c_void_p, c_int]
    lib.mi3pt_set_pipelining.argtypes = [c_void_p, c_int]
    lib.mi3pt_set_present_mode.argtypes = [c_void_p, c_int]
    for name in ("mi3pt_upload_triangles", "mi3pt_upload_materials", "mi3pt_upload_bvh"):
        getattr(lib, name).argtypes = [c_void_p, c_void_p, c_size_t]
    for name in ("mi3pt_upload_environment", "mi3pt_upload_environment_cdf"):
        getattr(lib, name).argtypes = [c_void_p, c_void_p, c_int, c_int]
    lib.mi3pt_resize.argtypes = [c_void_p, c_int, c_int]
    lib.mi3pt_reset.argtypes = [c_void_p]
    lib.mi3pt_set_uniforms.argtypes = [c_void_p, c_int, c_void_p, c_size_t]
    lib.mi3pt_submit.argtypes = [c_void_p, ctypes.c_uint]
    lib.mi3pt_submit_frames.argtypes = [c_void_p, ctypes.c_uint, ctypes.c_uint32]
    lib.mi3pt_sync.argtypes = [c_void_p]
    lib.mi3pt_flush.argtypes = [c_void_p]
    lib.mi3pt_read_texture.argtypes = [c_void_p, c_int, c_void_p, c_size_t]
    lib.mi3pt_read_canvas_rgba8.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.mi3pt_write_texture.argtypes = [c_void_p, c_int, c_void_p, c_size_t]
    lib.mi3pt_accumulation_device_ptr.argtypes = [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t)]
    lib.mi3pt_bind_accumulation.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.mi3pt_enable_timing.argtypes = [c_void_p, c_int]
    lib.mi3pt_pass_time_us.argtypes = [c_void_p, c_int, ctypes.POINTER(ctypes.c_float)]
    lib.mi3pt_raytrace_launch_stats.argtypes = [c_void_p, c_int, ctypes.POINTER(ctypes.c_double),
                                                ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.mi3pt_raytrace_launch_span.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_double)]
    lib.mi3pt_batch_capacity.argtypes = [c_void_p, ctypes.POINTER(c_int)]
    lib.mi3pt_debug_active_variant.argtypes = [c_void_p, ctypes.POINTER(c_int)]
    lib.mi3pt_debug_last_launch.argtypes = [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    lib.mi3pt_create_group.argtypes = [ctypes.POINTER(c_int), c_int, c_int, ctypes.POINTER(c_void_p)]
    lib.mi3pt_group_size.argtypes = [c_void_p, ctypes.POINTER(c_int)]
    lib.mi3pt_group_member.argtypes = [c_void_p, c_int, ctypes.POINTER(c_void_p)]
    lib.mi3pt_debug_set_option.argtypes = [c_void_p, c_int, c_int]
    lib.mi3pt_debug_get_option.argtypes = [c_void_p, c_int, ctypes.POINTER(c_int)]
    lib.mi3pt_debug_set_packet_layout.argtypes = [c_void_p, c_int]
    lib.mi3pt_get_counters.argtypes = [c_void_p, c_void_p]
    lib.mi3pt_reset_counters.argtypes = [c_void_p]
    lib.mi3pt_debug_intersect.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    lib.mi3pt_debug_math.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t]
    lib.mi3pt_device_build_bvh.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]
    lib.mi3pt_debug_wave_times.argtypes = [c_void_p, c_int, c_void_p, c_size_t, ctypes.POINTER(c_size_t)]
    lib.mi3pt_host_build_bvh.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, ctypes.POINTER(c_size_t), c_int]
    lib.mi3pt_host_build_bvh_f64.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, ctypes.POINTER(c_size_t), c_int]
    lib.mi3pt_host_env_cdf.argtypes = [c_void_p, c_int, c_int, c_void_p]
    lib.mi3pt_host_eight_wide_check.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p]
    lib.mi3pt_device_count.argtypes = [ctypes.POINTER(c_int)]
    lib.mi3pt_device_name.argtypes = [c_int, ctypes.c_char_p, c_size_t]
    lib.mi3pt_tile_local_rows.argtypes = [c_int, c_int, c_int, c_int]
    lib.mi3pt_tile_global_row.argtypes = [c_int, c_int, c_int, c_int]
    lib.mi3pt_tile_owner.argtypes = [c_int, c_int, c_int]
    if path == LIB_PATH:
        _lib = lib
    return lib


def _check(lib, rc):
    if rc != 0:
        raise Mi3ptError(rc, lib.mi3pt_last_error().decode("utf-8", "replace"))


def _ptr(arr):
    return arr.ctypes.data_as(ctypes.c_void_p)


def device_count():
    lib = load_library()
    n = ctypes.c_int(0)
    _check(lib, lib.mi3pt_device_count(ctypes.byref(n)))
    return n.value


def device_name(device=0):
    lib = load_library()
    buf = ctypes.create_string_buffer(256)
    _check(lib, lib.mi3pt_device_name(device, buf, 256))
    return buf.value.decode()


def last_error():
    """Text of this thread's most recent error -- or warning (the launch gate's host-side release returns MI3PT_OK and leaves one)."""
    return load_library().mi3pt_last_error().decode(errors="replace")


def tile_local_rows(height, rank, nranks, block_rows):
    return load_library().mi3pt_tile_local_rows(height, rank, nranks, block_rows)


def tile_global_row(local_row, rank, nranks, block_rows):
    """The image row that local row `local_row` of `rank`'s compact texture holds (mi3pt_set_tile's deal)."""
    return load_library().mi3pt_tile_global_row(local_row, rank, nranks, block_rows)


def tile_owner(y, nranks, block_rows):
    return load_library().mi3pt_tile_owner(y, nranks, block_rows)


# ---- host-side scene compile (no device) ----

def host_build_bvh(triangles, nthreads=0):
    """triangles: structured array (layout.TRIANGLE) or raw bytes -> BVH_NODE array."""
    from . import layout
    lib = load_library()
    tri = np.ascontiguousarray(triangles)
    n = tri.nbytes // 112
    nodes = np.zeros(max(2 * n - 1, 1), layout.BVH_NODE)
    count = ctypes.c_size_t(0)
    _check(lib, lib.mi3pt_host_build_bvh(_ptr(tri), n, _ptr(nodes), nodes.nbytes, ctypes.byref(count), nthreads))
    return nodes[:count.value]


def host_build_bvh_f64(positions, nthreads=0):
    """positions: (n, 3, 3) float64 world-space triangle vertices."""
    from . import layout
    lib = load_library()
    pos = np.ascontiguousarray(positions, np.float64)
    n = pos.size // 9
    nodes = np.zeros(max(2 * n - 1, 1), layout.BVH_NODE)
    count = ctypes.c_size_t(0)
    _check(lib, lib.mi3pt_host_build_bvh_f64(_ptr(pos), n, _ptr(nodes), nodes.nbytes, ctypes.byref(count), nthreads))
    return nodes[:count.value]


def host_eight_wide_check(nodes, triangles, greedy=False):
    """mi3pt_host_eight_wide_check: builds kernel variant 14's packets on the host and checks them independently of the builder;
    dict(packets, records, levels, children_per_packet, leaves, offered)."""
    lib = load_library()
    nd, tr = np.ascontiguousarray(nodes), np.ascontiguousarray(triangles)
    out = np.zeros(6, np.uint64)
    _check(lib, lib.mi3pt_host_eight_wide_check(_ptr(nd), nd.nbytes, _ptr(tr), tr.nbytes, 1 if greedy else 0, _ptr(out)))
    return {"packets": int(out[0]), "records": int(out[1]), "levels": int(out[2]), "children_per_packet": out[3] / 1000.0,
            "leaves": int(out[4]), "offered": bool(out[5])}


def host_env_cdf(rgba):
    lib = load_library()
    env = np.ascontiguousarray(rgba, np.float32)
    h, w = env.shape[0], env.shape[1]
    out = np.empty((h, w, 4), np.float32)
    _check(lib, lib.mi3pt_host_env_cdf(_ptr(env), w, h, _ptr(out)))
    return out


class Context:
    """One mi3pt_ctx: a HIP device + stream + the path tracer's device resources."""

    def __init__(self, device=0, devices=None, block_rows=8):
        """device: one GPU.  devices = [d0, d1, ...]: a device group (mi3pt_create_group) -- the same methods, whole
        images in and out, the image's 8-row blocks dealt to one member context per listed device."""
        self.lib = load_library()
        handle = ctypes.c_void_p()
        self.group_size = 1
        self._group_block_rows = block_rows
        if devices is not None:
            arr = (ctypes.c_int * len(devices))(*devices)
            _check(self.lib, self.lib.mi3pt_create_group(arr, len(devices), block_rows, ctypes.byref(handle)))
            self.group_size = len(devices)
        else:
            _check(self.lib, self.lib.mi3pt_create(device, ctypes.byref(handle)))
        self.handle = handle
        self.width = self.height = 0
        self.local_rows = 0
        self._tile = (0, 1, 8)
        self._next_tile = (0, 1, 8)

    def close(self):
        if self.handle and not getattr(self, "_borrowed", False):
            self.lib.mi3pt_destroy(self.handle)
        self.handle = None

    def member(self, index):
        """A member context of a device group (index -1: the presenting context), owned by the group."""
        h = ctypes.c_void_p()
        self._c(self.lib.mi3pt_group_member(self.handle, index, ctypes.byref(h)))
        m = Context.__new__(Context)
        m.lib, m.handle, m._borrowed, m.group_size = self.lib, h, True, 1
        m.width, m.height = self.width, self.height
        n = self.group_size
        m._group_block_rows = 8
        m._tile = m._next_tile = (0, 1, 8) if index < 0 or n == 1 else (index, n, self._group_block_rows)
        m.local_rows = tile_local_rows(self.height, *m._tile) if self.height else 0
        return m

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _c(self, rc):
        _check(self.lib, rc)

    def set_stream(self, stream_ptr):
        self._c(self.lib.mi3pt_set_stream(self.handle, stream_ptr))

    def set_storage(self, storage):
        self._c(self.lib.mi3pt_set_storage(self.handle, storage))

    def set_kernel_variant(self, variant):
        self._c(self.lib.mi3pt_set_kernel_variant(self.handle, variant))

    def set_env_sampling(self, enabled):
        """Run the reference's dormant environment importance sampling (raytrace.wgsl:398-404 un-commented)."""
        self._c(self.lib.mi3pt_set_env_sampling(self.handle, int(bool(enabled))))

    def set_pipelining(self, enabled):
        self._c(self.lib.mi3pt_set_pipelining(self.handle, int(enabled)))

    def set_present_mode(self, mode):
        self._c(self.lib.mi3pt_set_present_mode(self.handle, int(mode)))

    def set_tile(self, rank, nranks, block_rows=8):
        self._c(self.lib.mi3pt_set_tile(self.handle, rank, nranks, block_rows))
        self._next_tile = (rank, nranks, block_rows)

    def upload_triangles(self, tris):
        a = np.ascontiguousarray(tris)
        self._c(self.lib.mi3pt_upload_triangles(self.handle, _ptr(a), a.nbytes))
        self._ntris_uploaded = a.nbytes // 112

    def upload_materials(self, mats):
        a = np.ascontiguousarray(mats)
        self._c(self.lib.mi3pt_upload_materials(self.handle, _ptr(a), a.nbytes))

    def upload_bvh(self, nodes):
        a = np.ascontiguousarray(nodes)
        self._c(self.lib.mi3pt_upload_bvh(self.handle, _ptr(a), a.nbytes))

    def upload_environment(self, rgba):
        a = np.ascontiguousarray(rgba, np.float32)
        self._c(self.lib.mi3pt_upload_environment(self.handle, _ptr(a), a.shape[1], a.shape[0]))

    def upload_environment_cdf(self, rgba):
        a = np.ascontiguousarray(rgba, np.float32)
        self._c(self.lib.mi3pt_upload_environment_cdf(self.handle, _ptr(a), a.shape[1], a.shape[0]))

    def resize(self, width, height):
        self._c(self.lib.mi3pt_resize(self.handle, width, height))
        self.width, self.height = width, height
        self._tile = self._next_tile
        self.local_rows = tile_local_rows(height, *self._tile)

    def reset(self):
        self._c(self.lib.mi3pt_reset(self.handle))

    def set_uniforms(self, which, data):
        b = data if isinstance(data, (bytes, bytearray)) else data.tobytes()
        self._c(self.lib.mi3pt_set_uniforms(self.handle, which, b, len(b)))

    def submit(self, mask):
        self._c(self.lib.mi3pt_submit(self.handle, mask))

    def submit_frames(self, mask, count):
        self._c(self.lib.mi3pt_submit_frames(self.handle, mask, count))

    def flush(self):
        self._c(self.lib.mi3pt_flush(self.handle))

    def sync(self):
        self._c(self.lib.mi3pt_sync(self.handle))

    def read_texture(self, which):
        rows = self.height if which == TEX_CANVAS else self.local_rows
        out = np.empty((rows, self.width, 4), np.float32)
        self._c(self.lib.mi3pt_read_texture(self.handle, which, _ptr(out), out.size))
        return out

    def write_texture(self, which, image):
        a = np.ascontiguousarray(image, np.float32)
        self._c(self.lib.mi3pt_write_texture(self.handle, which, _ptr(a), a.size))

    def read_canvas_rgba8(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._c(self.lib.mi3pt_read_canvas_rgba8(self.handle, _ptr(out), out.nbytes))
        return out

    def accumulation_device_ptr(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._c(self.lib.mi3pt_accumulation_device_ptr(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def bind_accumulation(self, dev_ptr, nbytes):
        self._c(self.lib.mi3pt_bind_accumulation(self.handle, dev_ptr, nbytes))

    def enable_timing(self, enabled=True):
        self._c(self.lib.mi3pt_enable_timing(self.handle, int(enabled)))

    def pass_time_us(self, which):
        v = ctypes.c_float()
        self._c(self.lib.mi3pt_pass_time_us(self.handle, which, ctypes.byref(v)))
        return v.value

    def raytrace_launch_stats(self, reset=False):
        """(total GPU ms, launches, frames) of the batched raytrace kernel since the last reset."""
        ms, n, f = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        self._c(self.lib.mi3pt_raytrace_launch_stats(self.handle, int(reset), ctypes.byref(ms), ctypes.byref(n),
                                                     ctypes.byref(f)))
        return ms.value, n.value, f.value

    def set_packet_layout(self, layout):
        self._c(self.lib.mi3pt_debug_set_packet_layout(self.handle, int(layout)))

    def set_option(self, option, value):
        """Scheduling options (OPT_*): how the work is cut into launches / steps / jobs; never what is computed."""
        self._c(self.lib.mi3pt_debug_set_option(self.handle, int(option), int(value)))

    def get_option(self, option):
        v = ctypes.c_int(0)
        self._c(self.lib.mi3pt_debug_get_option(self.handle, int(option), ctypes.byref(v)))
        return v.value

    def active_variant(self):
        v = ctypes.c_int()
        self._c(self.lib.mi3pt_debug_active_variant(self.handle, ctypes.byref(v)))
        return v.value

    def last_launch(self):
        """The kernel the most recent raytrace launch ran: dict(kind, variant, lean, workgroups) -- mi3pt_debug_last_launch."""
        k, v, l, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._c(self.lib.mi3pt_debug_last_launch(self.handle, ctypes.byref(k), ctypes.byref(v), ctypes.byref(l), ctypes.byref(w)))
        b = self.get_option(OPT_LAST_BUILD)
        return {"kind": k.value, "variant": v.value, "lean": bool(l.value), "workgroups": w.value,
                "waves_per_simd": b & 0xff, "ymax": bool(b & 0x100), "walk_min": (b >> 16) & 0xff}

    def batch_capacity(self):
        n = ctypes.c_int()
        self._c(self.lib.mi3pt_batch_capacity(self.handle, ctypes.byref(n)))
        return n.value

    def raytrace_launch_span(self):
        """GPU-clock ms from the start of the first to the end of the last batched launch since the reset."""
        ms = ctypes.c_double()
        self._c(self.lib.mi3pt_raytrace_launch_span(self.handle, ctypes.byref(ms)))
        return ms.value

    def counters(self):
        out = np.zeros(8, np.uint64)
        self._c(self.lib.mi3pt_get_counters(self.handle, _ptr(out)))
        return dict(zip(COUNTER_NAMES, (int(x) for x in out)))

    def reset_counters(self):
        self._c(self.lib.mi3pt_reset_counters(self.handle))

    def debug_intersect(self, rays):
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.empty((len(r), 12), np.float32)
        self._c(self.lib.mi3pt_debug_intersect(self.handle, _ptr(r), len(r), _ptr(out)))
        return out

    def enable_wave_times(self, enabled=True):
        self._c(self.lib.mi3pt_debug_wave_times(self.handle, int(enabled), None, 0, None))

    def wave_times(self):
        out = np.zeros((6144, 16), np.uint64)
        n = ctypes.c_size_t()
        self._c(self.lib.mi3pt_debug_wave_times(self.handle, 1, _ptr(out), len(out), ctypes.byref(n)))
        return out[:n.value]

    def device_build_bvh(self):
        """Linear BVH over the uploaded triangles, built on the device: (nodes, build_ms).  An alternative
        to the reference's SAH tree (host_build_bvh_f64), in the same 48-byte records."""
        n = ctypes.c_size_t()
        ms = ctypes.c_float()
        from . import layout
        ntris = getattr(self, "_ntris_uploaded", 0)
        if ntris == 0:
            raise Mi3ptError(4, "no triangles uploaded")
        nodes = np.zeros(2 * ntris - 1, layout.BVH_NODE)
        self._c(self.lib.mi3pt_device_build_bvh(self.handle, _ptr(nodes), nodes.nbytes, ctypes.byref(n), ctypes.byref(ms)))
        return nodes[: n.value], ms.value

    def debug_math(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        out = np.empty_like(a)
        bb = np.ascontiguousarray(b, np.float32) if b is not None else None
        self._c(self.lib.mi3pt_debug_math(self.handle, fn, _ptr(a), _ptr(bb) if bb is not None else None,
                                          _ptr(out), a.size))
        return out
