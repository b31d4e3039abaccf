"""ctypes wrapper around oracle/libptoracle.so -- TEST INFRASTRUCTURE, not product.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this.
PARITY UNPINNED (see pt_oracle.c): the reference has no golden vectors and cannot run
here, so this oracle is pinned by hand-derived known-answer tests only.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libptoracle.so")


class _Scene(ctypes.Structure):
    _fields_ = [("tris", ctypes.c_void_p), ("ntris", ctypes.c_uint32),
                ("mats", ctypes.c_void_p), ("nmats", ctypes.c_uint32),
                ("nodes", ctypes.c_void_p), ("nnodes", ctypes.c_uint32),
                ("env", ctypes.c_void_p), ("env_w", ctypes.c_int32), ("env_h", ctypes.c_int32),
                ("cdf", ctypes.c_void_p), ("env_sampling", ctypes.c_int32)]


_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", HERE] + (["-B"] if force else []))


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        L.orc_tile_local_rows.argtypes = [ctypes.c_int] * 4
        L.orc_rand_sequence.restype = ctypes.c_uint32
        L.orc_rand_sequence.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        L.orc_ray_aabb.argtypes = [ctypes.c_void_p] * 4
        L.orc_ray_triangle.argtypes = [ctypes.c_void_p] * 4
        L.orc_ray_scene.argtypes = [ctypes.c_void_p] * 5
        L.orc_camera_ray.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
        L.orc_env_uv.argtypes = [ctypes.c_void_p] * 3
        L.orc_sample_env.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
        L.orc_sample_repeat.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_void_p]
        L.orc_math.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_raytrace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_accumulate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_fullscreen.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
        L.orc_set_num_threads(default_threads())
    return _lib


def cpu_quota():
    """CPUs this process may actually use: the cgroup CPU quota (v2 `cpu.max`, v1 `cpu.cfs_quota_us`) where one is
    set, else None.  OpenMP only sees the affinity mask: on a GPU box that shows every core of the host (256) while the
    container's share is 16 -- 256 threads on 16 cores ran the oracle at a tenth of its speed (round-4 verdict)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, int(float(quota) / float(period) + 0.5))
    except (OSError, ValueError):
        pass
    for base in ("/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
        try:
            with open(base + "/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open(base + "/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                return max(1, int(quota / period + 0.5))
        except (OSError, ValueError):
            pass
    return None


def default_threads():
    """Thread count of the oracle's passes unless a caller sets another: min(cgroup quota, affinity mask); with no
    quota, the affinity mask capped at PT_ORACLE_MAX_THREADS (default 32: the GPU boxes' share is 16 and past a
    host's real share the oracle only gets slower)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    quota = cpu_quota()
    if quota is not None:
        return max(1, min(quota, avail))
    return max(1, min(avail, int(os.environ.get("PT_ORACLE_MAX_THREADS", "32"))))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


COUNTER_NAMES = ("rays", "box_tests", "tri_tests", "hits", "misses", "stack_overflows", "pixels", "reserved")


class OracleScene:
    """Holds the byte buffers (reference layouts) alive for the C side."""

    def __init__(self, triangles, materials, nodes, env=None, cdf=None, env_sampling=False):
        self.tris = np.ascontiguousarray(triangles)
        self.mats = np.ascontiguousarray(materials)
        self.nodes = np.ascontiguousarray(nodes) if nodes is not None else np.zeros(0, np.uint8)
        if env is None:
            env = np.zeros((512, 1024, 4), np.float32)
        self.env = np.ascontiguousarray(env, np.float32)
        self.c = _Scene(_p(self.tris), self.tris.nbytes // 112, _p(self.mats), self.mats.nbytes // 64,
                        _p(self.nodes), self.nodes.nbytes // 48, _p(self.env),
                        self.env.shape[1], self.env.shape[0], None, 0)
        self.cdf = None
        if cdf is not None:
            self.cdf = np.ascontiguousarray(cdf, np.float32)
            assert self.cdf.shape == self.env.shape
            self.c.cdf = self.cdf.ctypes.data
            self.c.env_sampling = 1 if env_sampling else 0


def tile_local_rows(h, rank, nranks, block_rows):
    return lib().orc_tile_local_rows(h, rank, nranks, block_rows)


def set_num_threads(n):
    """OpenMP threads of the following passes (bench.py's thread-scaling baseline)."""
    lib().orc_set_num_threads(int(n))


def max_threads():
    return int(lib().orc_max_threads())


def num_procs():
    """Processors OpenMP sees (its affinity mask; a CPU quota imposed from outside is not visible here)."""
    return int(lib().orc_num_procs())


def raytrace(scene, uniforms96, tex_w, tex_h, rank=0, nranks=1, block_rows=8, store_f16=False, out=None):
    """One raytrace pass.  Returns (image[local_rows, tex_w, 4], counters dict)."""
    rows = tile_local_rows(tex_h, rank, nranks, block_rows)
    if out is None:
        out = np.zeros((rows, tex_w, 4), np.float32)
    cnt = np.zeros(8, np.uint64)
    u = np.frombuffer(bytes(uniforms96), np.uint8).copy()
    lib().orc_raytrace(ctypes.byref(scene.c), _p(u), tex_w, tex_h, rank, nranks, block_rows,
                       int(store_f16), _p(out), _p(cnt))
    return out, dict(zip(COUNTER_NAMES, (int(x) for x in cnt)))


def accumulate(uniforms16, tex_w, tex_h, inp, prev, rank=0, nranks=1, block_rows=8, store_f16=False):
    """accumulate pass; returns the new accumulation image (prev is not modified)."""
    out = prev.copy()
    u = np.frombuffer(bytes(uniforms16), np.uint8).copy()
    lib().orc_accumulate(_p(u), tex_w, tex_h, rank, nranks, block_rows, int(store_f16),
                         _p(np.ascontiguousarray(inp)), _p(np.ascontiguousarray(prev)), _p(out))
    return out


def fullscreen(uniforms24, tex, canvas_w=None, canvas_h=None):
    """fullscreen pass; returns (float image [H, W, 4], rgba8 image)."""
    tex = np.ascontiguousarray(tex, np.float32)
    th, tw = tex.shape[0], tex.shape[1]
    cw, ch = canvas_w or tw, canvas_h or th
    of = np.zeros((ch, cw, 4), np.float32)
    o8 = np.zeros((ch, cw, 4), np.uint8)
    u = np.frombuffer(bytes(uniforms24), np.uint8).copy()
    lib().orc_fullscreen(_p(u), _p(tex), tw, th, cw, ch, _p(of), _p(o8))
    return of, o8


def math_fn(fn, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    lib().orc_math(fn, _p(a), _p(bb), _p(out), a.size)
    return out


def rand_sequence(seed, n):
    out = np.empty(n, np.float32)
    final = lib().orc_rand_sequence(ctypes.c_uint32(seed & 0xFFFFFFFF), n, _p(out))
    return out, final


def ray_aabb(o, d, bmin, bmax):
    f = lambda v: np.ascontiguousarray(v, np.float32)
    o, d, bmin, bmax = f(o), f(d), f(bmin), f(bmax)
    return bool(lib().orc_ray_aabb(_p(o), _p(d), _p(bmin), _p(bmax)))


def ray_triangle(o, d, tri112):
    o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
    t = np.ascontiguousarray(tri112)
    out = np.zeros(9, np.float32)
    lib().orc_ray_triangle(_p(o), _p(d), _p(t), _p(out))
    return out


def ray_scene(scene, o, d):
    o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
    out = np.zeros(9, np.float32)
    cnt = np.zeros(8, np.uint64)
    lib().orc_ray_scene(ctypes.byref(scene.c), _p(o), _p(d), _p(out), _p(cnt))
    return out, dict(zip(COUNTER_NAMES, (int(x) for x in cnt)))


def camera_ray(uniforms96, uvx, uvy):
    u = np.frombuffer(bytes(uniforms96), np.uint8).copy()
    out = np.zeros(6, np.float32)
    lib().orc_camera_ray(_p(u), uvx, uvy, _p(out))
    return out


def env_uv(uniforms96, direction):
    u = np.frombuffer(bytes(uniforms96), np.uint8).copy()
    d = np.ascontiguousarray(direction, np.float32)
    out = np.zeros(2, np.float32)
    lib().orc_env_uv(_p(u), _p(d), _p(out))
    return out


def sample_env(scene, u, v):
    out = np.zeros(3, np.float32)
    lib().orc_sample_env(ctypes.byref(scene.c), u, v, _p(out))
    return out


def sample_repeat(tex, u, v):
    tex = np.ascontiguousarray(tex, np.float32)
    out = np.zeros(4, np.float32)
    lib().orc_sample_repeat(_p(tex), tex.shape[1], tex.shape[0], u, v, _p(out))
    return out
