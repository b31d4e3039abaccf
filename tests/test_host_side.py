"""CPU tests of the host side: byte layouts, the BVH builder (tree identity with a literal
transcription of the reference), the environment CDF, scene generators, the C-ABI
library's exported symbols, and the Renderer state machine.  No GPU needed."""
import ctypes
import math
import os
import re
import subprocess

import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi, layout, scenes
from mi3pt_host.renderer import RaytracingCamera, RaytracingScene, Renderer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- layouts

def test_struct_sizes_and_offsets():
    assert layout.TRIANGLE.itemsize == 112 and layout.BVH_NODE.itemsize == 48 and layout.MATERIAL.itemsize == 64
    assert layout.RAYTRACE_UNIFORMS.itemsize == 96 and layout.ACCUMULATE_UNIFORMS.itemsize == 16
    assert layout.FULLSCREEN_UNIFORMS.itemsize == 24
    off = lambda dt, name: dt.fields[name][1]
    assert [off(layout.TRIANGLE, n) for n in layout.TRIANGLE.names] == [0, 16, 32, 48, 64, 80, 92, 96]
    assert [off(layout.BVH_NODE, n) for n in layout.BVH_NODE.names] == [0, 16, 28, 32, 36, 40]
    assert [off(layout.MATERIAL, n) for n in layout.MATERIAL.names] == [0, 16, 28, 32, 48, 60]
    assert [off(layout.RAYTRACE_UNIFORMS, n) for n in layout.RAYTRACE_UNIFORMS.names] == \
        [0, 8, 12, 16, 20, 32, 48, 60, 64, 68, 80, 84]
    assert [off(layout.ACCUMULATE_UNIFORMS, n) for n in layout.ACCUMULATE_UNIFORMS.names] == [0, 8, 12]
    assert [off(layout.FULLSCREEN_UNIFORMS, n) for n in layout.FULLSCREEN_UNIFORMS.names] == [0, 8, 12, 16, 20]


def test_header_constants_match_layouts():
    hdr = open(os.path.join(ROOT, "include", "mi3pt.h")).read()
    for name, want in (("TRIANGLE_STRIDE", 112), ("BVHNODE_STRIDE", 48), ("MATERIAL_STRIDE", 64),
                       ("RAYTRACE_UNIFORMS_SIZE", 96), ("ACCUMULATE_UNIFORMS_SIZE", 16),
                       ("FULLSCREEN_UNIFORMS_SIZE", 24), ("ENV_WIDTH", 1024), ("ENV_HEIGHT", 512)):
        assert int(re.search(rf"#define MI3PT_{name} (\d+)", hdr).group(1)) == want


def test_uniform_block_partial_set_and_u32_truncation():
    u = layout.UniformBlock(layout.RAYTRACE_UNIFORMS)
    u.set({"maxBounces": 4, "envMapIntensity": 1.5})
    u.set({"camera": {"fov": 45.0, "position": (0, 1, 4)}, "frame": 7, "unknownKey": 3})
    assert u.get("maxBounces") == 4 and u.get("envMapIntensity") == 1.5 and u.get("frame") == 7
    assert tuple(u.get("camera.position")) == (0, 1, 4)
    raw = u.tobytes()
    assert len(raw) == 96 and np.frombuffer(raw, "<i4", 1, 16)[0] == 4
    assert np.frombuffer(raw, "<f4", 1, 60)[0] == 45.0
    a = layout.UniformBlock(layout.ACCUMULATE_UNIFORMS)
    a.set({"resolution": [479.75, 269.99], "frame": 3})         # float -> u32 store truncates
    assert tuple(a.get("resolution")) == (479, 269)


# ---------------------------------------------------------------- the C ABI library

def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mi3pt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"#ifdef MI3PT_EXPERIMENTS.*?#endif", "", hdr, flags=re.S)       # (declared for the experiment build only)
    return sorted(set(re.findall(r"\b(mi3pt_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(built):
    declared = _declared_symbols()
    assert sorted(capi.SYMBOLS) == declared
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libmi3pt.so does not export {name}"
    assert lib.mi3pt_abi_version() == 4
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (mi3pt_\w+)", out))
    assert exported == set(declared)
    assert "orc_" not in out                    # no oracle / CPU render path inside the product


def test_release_library_reads_no_knobs(built):
    """The release object reads no MI3PT_* environment variable (round-2 finding: an environment variable could void
    the bit-exactness guarantee of a drop-in library): no knob name is in the binary, and the only environment names it
    knows are the profiler's (the launch gate switches itself off under rocprofv3 --pmc).  The scheduling options are
    an explicit call, mi3pt_debug_set_option; the two switches that change what is computed exist only in the
    -DMI3PT_EXPERIMENTS build (make -C csrc experiments)."""
    blob = open(capi.LIB_PATH, "rb").read()
    src = open(os.path.join(ROOT, "webgpu-pathtracer_amd", "csrc", "pt_context.hip")).read()
    knobs = set(re.findall(r'"(MI3PT_[A-Z_]+)"', src))
    assert {"MI3PT_CULL_SCALE", "MI3PT_FORCE_SLOW_SLAB", "MI3PT_GATE", "MI3PT_BATCH", "MI3PT_JOB_CHUNK"} <= knobs
    for k in sorted(knobs):
        assert k.encode() + b"\0" not in blob, f"{k} is in the release library"
    # every getenv of an MI3PT_ name sits inside an #ifdef MI3PT_EXPERIMENTS block
    depth, guarded = 0, []
    for line in src.split("\n"):
        t = line.strip()
        if t.startswith("#ifdef MI3PT_EXPERIMENTS"):
            depth += 1
        elif t.startswith("#endif") and depth:
            depth -= 1
        if "getenv" in line and "MI3PT_" in line or ("getenv(eo.name)" in line):
            guarded.append(depth > 0)
    assert guarded and all(guarded)


def test_product_sources_do_not_reference_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "webgpu-pathtracer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".js", ".cc", ".ts")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "pt_oracle" not in text and "libptoracle" not in text, os.path.join(base, f)


def test_no_device_means_error_not_fallback(built):
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(capi.Mi3ptError) as e:
        capi.Context(0)
    assert e.value.code == 2 and "HIP device not found" in e.value.message
    assert Renderer.diagnostic() == {"supported": False}
    with pytest.raises(capi.Mi3ptError):
        Renderer.create()
    with pytest.raises(capi.Mi3ptError) as e:          # a device group fails the same way: no member can be created
        capi.Context(devices=[0, 1])
    assert e.value.code == 2 and "HIP device not found" in e.value.message
    with pytest.raises(capi.Mi3ptError) as e:
        capi.Context(devices=[])
    assert e.value.code == 1


def test_host_argument_validation(built):
    lib = capi.load_library()
    assert lib.mi3pt_host_build_bvh(None, 0, None, 0, None, 1) == 1
    assert b"Input nodes array is empty" in lib.mi3pt_last_error()     # raytrace.ts:563-565
    tri = np.zeros(3, layout.TRIANGLE)
    small = np.zeros(2, layout.BVH_NODE)
    n = ctypes.c_size_t()
    assert lib.mi3pt_host_build_bvh(tri.ctypes.data, 3, small.ctypes.data, small.nbytes, ctypes.byref(n), 1) == 1
    assert capi.tile_local_rows(70, 1, 3, 5) == len([y for y in range(70) if (y // 5) % 3 == 1])
    assert capi.tile_local_rows(70, 3, 3, 5) == -1


# ---------------------------------------------------------------- BVH builder

def _random_triangles(rng, n, kind):
    if kind == "blob":
        c = rng.normal(size=(n, 1, 3)) * 2.0
        return c + rng.normal(size=(n, 3, 3)) * 0.3
    if kind == "grid":              # many equal centres / ties: exercises the stable sort and strict '<'
        c = rng.integers(0, 4, size=(n, 1, 3)).astype(float)
        return c + np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], float)
    if kind == "flat":              # all on a plane: the axis rule picks y whenever x <= y
        p = rng.normal(size=(n, 3, 3))
        p[..., 1] = 0.0
        return p
    p = rng.normal(size=(n, 3, 3))  # "long-z": z is the longest axis but x <= y -> 'y' is chosen
    p[..., 2] *= 10.0
    p[..., 0] *= 0.5
    return p


@pytest.mark.parametrize("kind", ["blob", "grid", "flat", "long-z"])
@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 257])
def test_builder_matches_literal_transcription(built, kind, n):
    import bvh_literal
    rng = np.random.default_rng(n * 31 + len(kind))
    pos = _random_triangles(rng, n, kind)
    got = capi.host_build_bvh_f64(pos, nthreads=3)
    want = bvh_literal.flatten_bvh(bvh_literal.build_bvh(pos.tolist()))
    assert len(got) == len(want) == 2 * n - 1
    for i, w in enumerate(want):
        g = got[i]
        assert (g["isLeaf"], g["left"], g["right"], g["triangleIndex"]) == \
            (w["isLeaf"], w["left"], w["right"], w["triangleIndex"]), f"node {i}"
        assert tuple(g["min"]) == tuple(np.float32(w["min"])) and tuple(g["max"]) == tuple(np.float32(w["max"]))


def test_builder_structure_on_the_demo_scene(demo):
    nodes = demo.nodes
    assert len(demo.triangles) == 1998 and len(nodes) == 3995
    internal = nodes["isLeaf"] != 1
    assert (nodes["right"][internal] == nodes["left"][internal] + 1).all()      # BFS adjacency
    assert (nodes["left"][internal] > np.nonzero(internal)[0]).all()             # children after parents
    assert (nodes["left"][~internal] == -1).all() and (nodes["triangleIndex"][internal] == -1).all()
    assert sorted(nodes["triangleIndex"][~internal]) == list(range(1998))        # one triangle per leaf
    # every child box lies inside its parent's box
    for i in np.nonzero(internal)[0][:500]:
        for c in (nodes["left"][i], nodes["right"][i]):
            assert (nodes["min"][c] >= nodes["min"][i]).all() and (nodes["max"][c] <= nodes["max"][i]).all()
    # The f32 entry point (positions re-read from the 112-B records) is a valid tree over the
    # same triangles, but NOT necessarily the same tree: the reference sorts and sweeps on the
    # JavaScript doubles (raytrace.ts:546-550), and near-ties split differently after fp32
    # rounding -- which is why the host passes doubles to mi3pt_host_build_bvh_f64.
    again = capi.host_build_bvh(demo.triangles, nthreads=2)
    assert len(again) == len(nodes)
    assert sorted(again["triangleIndex"][again["isLeaf"] == 1]) == list(range(1998))
    assert tuple(again[0]["min"]) == tuple(nodes[0]["min"]) and tuple(again[0]["max"]) == tuple(nodes[0]["max"])


def test_builder_is_deterministic_across_thread_counts(built):
    rng = np.random.default_rng(3)
    pos = _random_triangles(rng, 20000, "blob")
    a = capi.host_build_bvh_f64(pos, nthreads=1)
    b = capi.host_build_bvh_f64(pos, nthreads=8)
    assert a.tobytes() == b.tobytes()


# ---------------------------------------------------------------- environment CDF (renderer.ts:159-266)

def test_env_cdf_against_a_direct_float64_restatement(built):
    rng = np.random.default_rng(1)
    env = np.ones((512, 1024, 4), np.float32)
    env[..., :3] = rng.uniform(0.0, 4.0, size=(512, 1024, 3)).astype(np.float32)
    got = capi.host_env_cdf(env)
    lum = (0.2126 * env[..., 0].astype(np.float64) + 0.7152 * env[..., 1] + 0.0722 * env[..., 2]).astype(np.float32)
    wgt = np.sin((np.arange(512) + 0.5) / 512 * math.pi)
    weighted = (lum.astype(np.float64) * wgt[:, None]).astype(np.float32)
    rows = weighted.astype(np.float64).sum(1)
    marg = np.concatenate([[0.0], np.cumsum((rows.astype(np.float32) / rows.sum()).astype(np.float32).astype(np.float64))[:-1]])
    cond_src = (lum.astype(np.float64) / lum.astype(np.float64).sum(1, keepdims=True)).astype(np.float32).astype(np.float64)
    cond = np.concatenate([np.zeros((512, 1)), np.cumsum(cond_src, axis=1)[:, :-1]], axis=1)
    # (sin() differs by an ulp between libm and numpy, so fp32-close rather than equal)
    assert np.allclose(got[..., 2], weighted, rtol=3e-7, atol=0) and (got[..., 3] == 1).all()
    assert np.allclose(got[:, 0, 0], marg, rtol=1e-5, atol=1e-7) and (got[:, 5, 0] == got[:, 0, 0]).all()
    assert np.allclose(got[..., 1], cond, rtol=1e-5, atol=1e-7)
    assert got[0, 0, 0] == 0 and (got[:, 0, 1] == 0).all()               # exclusive prefixes
    assert 0.99 < got[-1, 0, 0] <= 1.0 and (np.diff(got[:, 0, 0]) >= 0).all()


# ---------------------------------------------------------------- scenes

def test_demo_scene_geometry_facts():
    sc = scenes.demo_scene()
    p = sc.positions
    assert len(p) == 2 + 12 + 1984
    plane, box, sphere = p[:2], p[2:14], p[14:]
    assert np.allclose(np.abs(plane[..., [0, 2]]).max(), 2.5) and np.abs(plane[..., 1]).max() < 1e-15
    assert np.allclose(box.reshape(-1, 3).min(0), (-0.4, 0.0, 0.1)) and np.allclose(box.reshape(-1, 3).max(0), (0.4, 0.8, 0.9))
    centre = np.array([0.0, 0.5, -0.5])
    assert np.allclose(np.linalg.norm(sphere.reshape(-1, 3) - centre, axis=1), 0.5, atol=1e-7)
    assert set(sc.material_index[:2]) == {0} and set(sc.material_index[2:14]) == {1} and set(sc.material_index[14:]) == {0}
    # plane normals point up after rotateX(-pi/2); box normals are axis aligned; sphere normals radial
    assert np.allclose(sc.normals[:2], (0, 1, 0), atol=1e-15)
    assert np.allclose(np.abs(sc.normals[2:14]).sum(-1), 1.0)
    assert np.allclose(sc.normals[14:], (sphere - centre) / 0.5, atol=1e-6)
    # first plane triangle: indices (a, b, d) = (0, 2, 1) of the 2x2 grid
    assert np.allclose(plane[0], [(-2.5, 0, -2.5), (-2.5, 0, 2.5), (2.5, 0, -2.5)], atol=1e-12)
    assert sc.materials[0]["metalness"] == 0.02 and sc.materials[1]["color"] == (1.0, 0.05, 0.05)
    m = sc.material_bytes
    assert m[0]["emissionStrength"] == 1.0 and tuple(m[0]["emissionColor"]) == (0, 0, 0)


def test_synthetic_scenes_are_seeded_and_sized():
    a, b = scenes.displaced_blob(24, seed=5), scenes.displaced_blob(24, seed=5)
    assert np.array_equal(a[0], b[0]) and len(a[1]) == 2 * 24 * 24 - 2 * 24
    env = scenes.synthetic_env()
    assert env.shape == (512, 1024, 4) and env.dtype == np.float32 and (env[..., 3] == 1).all()
    assert env[..., :3].max() > 20 and env[..., :3].min() >= 0            # sun disk + non-negative sky
    assert 2 * 660 * 660 - 2 * 660 + 2 == 869882                          # dragon_class_scene() size


# ---------------------------------------------------------------- Renderer state machine (renderer.ts:283-468)

class FakeContext:
    """Records what a Renderer sends to the C ABI (host-logic test only)."""

    def __init__(self):
        self.calls = []
        self.uniforms = {}

    def __getattr__(self, name):
        def record(*args):
            self.calls.append((name,) + args)
            if name == "set_uniforms":
                self.uniforms[args[0]] = args[1]
        return record


def _make_renderer(frames=3):
    r = Renderer(FakeContext())
    r.frames = frames
    r.scalingFactor = 1
    content = scenes.demo_scene()
    content.nodes = np.zeros(1, layout.BVH_NODE)        # builder not needed for the state machine
    scene = RaytracingScene(content, scenes.synthetic_env())
    scene.needsUpdate = True
    return r, scene, RaytracingCamera(45.0)


def test_renderer_frame_counter_semantics(monkeypatch):
    monkeypatch.setattr(capi, "host_env_cdf", lambda env: np.zeros_like(env))
    r, scene, cam = _make_renderer(frames=3)
    events = []
    for ev in ("start", "reset", "progress", "complete", "resize"):
        r.on(ev, lambda *a, ev=ev: events.append(ev))
    assert r.status == "idle"
    r.resize(64, 32)
    assert r.status == "sampling" and events[:3] == ["reset", "start", "resize"] and r.frame == 1
    frames_seen = []
    for _ in range(5):
        r.render(scene, cam)
        raw = r.ctx.uniforms[capi.PASS_RAYTRACE]
        frames_seen.append(int(np.frombuffer(raw, "<u4", 1, 12)[0]))
    # first sampled frame carries frame = 2, exactly `frames` dispatches run, then idle
    assert frames_seen == [2, 3, 4, 4, 4]
    submits = [c[1] for c in r.ctx.calls if c[0] == "submit"]
    assert submits == [7, 7, 7, 4, 4]                    # raytrace|accumulate|fullscreen, then present only
    assert r.status == "idle" and events.count("complete") == 1 and events.count("progress") == 3
    assert not scene.needsUpdate
    uploads = [c[0] for c in r.ctx.calls if c[0].startswith("upload_")]
    assert uploads == ["upload_environment", "upload_environment_cdf", "upload_bvh", "upload_triangles", "upload_materials"]
    acc = np.frombuffer(r.ctx.uniforms[capi.PASS_ACCUMULATE], "<u4")
    assert tuple(acc[:3]) == (64, 32, 4)
    # pause / start / reset
    r.reset()
    assert r.frame == 1 and r.status == "sampling"
    r.pause()
    r.render(scene, cam)
    assert r.frame == 1 and [c[1] for c in r.ctx.calls if c[0] == "submit"][-1] == 4
    r.start()
    r.render(scene, cam)
    assert r.frame == 2
    assert math.isclose(r.progress, 2 / 4)


def test_renderer_scaled_resolution_and_partial_uniforms(monkeypatch):
    monkeypatch.setattr(capi, "host_env_cdf", lambda env: np.zeros_like(env))
    r, scene, cam = _make_renderer()
    r.setUniforms("raytrace", {"maxBounces": 8, "envMapIntensity": 1.0})
    r.setUniforms("fullscreen", {"denoise": 1, "tonemapping": 1})
    r.resize(90, 50)
    r.scalingFactor = 0.25
    r.render(scene, cam)
    rt = np.frombuffer(r.ctx.uniforms[capi.PASS_RAYTRACE], np.uint8)
    assert tuple(np.frombuffer(rt, "<f4", 2, 0)) == (22.5, 12.5)         # float resolution, fractional
    assert np.frombuffer(rt, "<i4", 1, 16)[0] == 8                         # earlier partial set survived
    assert np.isclose(np.frombuffer(rt, "<f4", 1, 8)[0], 90 / 50)          # aspect is the canvas aspect
    acc = np.frombuffer(r.ctx.uniforms[capi.PASS_ACCUMULATE], "<u4")
    assert tuple(acc[:2]) == (22, 12)                                      # u32 truncation
    fs = r.ctx.uniforms[capi.PASS_FULLSCREEN]
    assert tuple(np.frombuffer(fs, "<f4", 2, 0)) == (90, 50) and np.frombuffer(fs, "<f4", 1, 12)[0] == 0.25
    assert tuple(np.frombuffer(fs, "<u4", 2, 16)) == (1, 1)
    # camera block: position (0,1,4), direction towards the origin
    d = np.frombuffer(rt, "<f4", 3, 48)
    assert np.allclose(d, np.array([0, -1, -4]) / math.sqrt(17), atol=1e-7)


def test_renderer_rejects_bad_environment():
    r, scene, cam = _make_renderer()
    with pytest.raises(capi.Mi3ptError, match="1024x512"):
        r.updateEnvironmentTexture(np.zeros((256, 512, 4), np.float32))
    with pytest.raises(capi.Mi3ptError, match="floating point"):
        r.updateEnvironmentTexture(np.zeros((512, 1024, 4), np.uint8))


def test_ticket_division_by_multiplication():
    """The job decode of the persistent kernel divides tickets (< 2^31) by launch-invariant divisors as
    floor(t m / 2^(31+s)), s = ceil(log2 d), m = floor(2^(31+s) / d) + 1 (pt_kernels.hip fast_div_of /
    fast_div): the identity behind it, on edge cases and a random sample, in exact integers."""
    import random
    rng = random.Random(7)
    ds = [1, 2, 3, 5, 7, 8, 240, 241, 255, 256, 257, 3840, 4050, 16320, 129600, 1036800, 2**20 - 1, 2**20, 2**20 + 1,
          2**30, 2**31 - 1] + [rng.randrange(1, 2**31) for _ in range(300)]
    for d in ds:
        s = 0 if d <= 1 else (d - 1).bit_length()
        sh = 31 + s
        m = (1 << sh) // d + 1
        assert m < 2**32
        ts = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2**31 - 1, 2**31 - d if d < 2**31 else 0] + [rng.randrange(0, 2**31) for _ in range(200)]
        for t in ts:
            if 0 <= t < 2**31:
                assert (t * m) >> sh == t // d, (t, d)


def test_tile_deal_helpers_agree_with_each_other_and_the_oracle(built, orc):
    """mi3pt_tile_global_row / mi3pt_tile_owner (ABI 3): the deal for hosts that de-interleave on their own, so that nobody
    re-implements mi3pt_set_tile's formula -- it changed once (round 4: back and forth) under an unchanged ABI number."""
    from mi3pt_host import tiles
    for h, n, b in ((70, 3, 5), (1080, 8, 8), (2160, 4, 8), (17, 5, 4), (64, 1, 8), (9, 2, 16)):
        seen = np.full(h, -1)
        for r in range(n):
            rows = tiles.local_rows_of(h, r, n, b)
            assert len(rows) == capi.tile_local_rows(h, r, n, b) == orc.tile_local_rows(h, r, n, b)
            assert [capi.tile_global_row(ly, r, n, b) for ly in range(len(rows))] == rows
            for y in rows:
                assert capi.tile_owner(y, n, b) == r
                seen[y] = r
        assert (seen >= 0).all()
    assert capi.tile_global_row(8, 0, 4, 8) == 56          # round 1 runs backwards: rank 0's second block is the round's last
    assert capi.tile_global_row(0, 4, 4, 8) == -1 and capi.tile_owner(-1, 4, 8) == -1 and capi.tile_owner(5, 0, 8) == -1




# ---------------------------------------------------------------- the eight-wide packets of kernel variant 14 (host-side build + self-check)

def test_eight_wide_packets_pass_their_host_side_check(built):
    """mi3pt_host_eight_wide_check builds kernel variant 14's packets as the scene analysis does and walks them independently of the
    builder: every leaf triangle reachable exactly once, every packet referenced once, records = their leaf's box and triangle, every
    decoded child box containing everything below it.  Here: the demo scene, random soups (ragged sizes), a two-triangle tree, the
    hand-made deep trees of tests/test_gpu_culling.py (24 packet levels: offered; 40: more than the walk's stack holds)."""
    import importlib.util
    demo = scenes.demo_scene()
    demo.build_bvh()
    r = capi.host_eight_wide_check(demo.nodes, demo.triangles)
    assert r["leaves"] == len(demo.triangles) == 1998 and r["offered"] and 2 <= r["levels"] <= 8
    assert 4.0 < r["children_per_packet"] <= 8.0 and r["records"] >= r["leaves"]
    g = capi.host_eight_wide_check(demo.nodes, demo.triangles, greedy=True)          # the greedy grouping (MI3PT_OPT_COLLAPSE = 0): emptier packets, more of them
    assert g["leaves"] == 1998 and g["packets"] > r["packets"] and g["children_per_packet"] < r["children_per_packet"]
    rng = np.random.default_rng(8)
    for n in (2, 3, 9, 64, 65, 1000, 4097):
        pos = rng.normal(size=(n, 3, 3)) * 0.1 + rng.normal(size=(n, 1, 3))
        sc = scenes.Scene(pos, np.tile(np.array([0.0, 0.0, 1.0]), (n, 3, 1)), np.zeros(n, int), [scenes.WHITE], f"soup-{n}")
        sc.build_bvh()
        for greedy in (False, True):
            r = capi.host_eight_wide_check(sc.nodes, sc.triangles, greedy)
            assert r["leaves"] == n and r["offered"] and r["packets"] >= 1, (n, r)
            assert r["packets"] <= max(1, n - 1) and r["records"] <= 8 * r["packets"] + 8
    spec = importlib.util.spec_from_file_location("culling_scenes", os.path.join(ROOT, "tests", "test_gpu_culling.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for depth, offered in ((6, True), (24, True), (40, False)):
        tris, _, nodes = mod._deep_eight_wide_scene(depth)
        r = capi.host_eight_wide_check(nodes, tris, greedy=True)          # (the greedy grouping follows the areas these trees are built around;
        assert r["leaves"] == len(tris) and r["offered"] is offered and depth + 1 <= r["levels"] <= depth + 3, (depth, r)
        o = capi.host_eight_wide_check(nodes, tris)                       # the optimal one packs them flat)
        assert o["leaves"] == len(tris) and o["offered"] and o["levels"] < r["levels"]
    # a one-triangle tree has no packet to build: refused with a reason, not a crash
    one = scenes.Scene(rng.normal(size=(1, 3, 3)), np.tile(np.array([0.0, 0.0, 1.0]), (1, 3, 1)), np.zeros(1, int), [scenes.WHITE], "one")
    one.build_bvh()
    with pytest.raises(capi.Mi3ptError):
        capi.host_eight_wide_check(one.nodes, one.triangles)
