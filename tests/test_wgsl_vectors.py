"""The C oracle (and on a GPU the HIP kernels) against vectors produced by EXECUTING the
reference's shader text (src/passes/shaders/*.wgsl) with oracle/wgsl_interp.py.

tests/golden/wgsl_vectors.npz holds inputs and outputs only (generated on a machine that has the
reference checkout by tests/golden/make_wgsl_vectors.py); this is what pins the oracle to the
reference's own code for the device part of the path.  Everything must match bit for bit: the
interpreter and the oracle implement the same pinned arithmetic, so any difference is
a transcription error in the restatement."""
import os

import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi, layout, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VECTORS = os.path.join(ROOT, "tests", "golden", "wgsl_vectors.npz")
REF_SHADERS = "/root/reference/src/passes/shaders"


@pytest.fixture(scope="module")
def vec():
    return np.load(VECTORS)


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)) | ((a == 0) & (b == 0))).all())


def first_diff(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    idx = np.argwhere(bad)
    return f"{bad.sum()} of {bad.size} differ; first at {tuple(idx[0])}: got {a[tuple(idx[0])]!r} want {b[tuple(idx[0])]!r}" if len(idx) else "equal"


# ---------------------------------------------------------------- raytrace.wgsl pieces

def test_rand_sequences(orc, vec):
    for seed, want, final in zip(vec["rand_seeds"], vec["rand_values"], vec["rand_final_seed"]):
        got, got_final = orc.rand_sequence(int(seed), len(want))
        assert bits_equal(got, want), first_diff(got, want)
        assert got_final == int(final)


def test_slab_test(orc, vec):
    got = [orc.ray_aabb(o, d, lo, hi) for o, d, lo, hi in zip(vec["aabb_o"], vec["aabb_d"], vec["aabb_min"], vec["aabb_max"])]
    assert np.array_equal(np.array(got, np.uint8), vec["aabb_hit"])
    assert 0.2 < vec["aabb_hit"].mean() < 0.8                      # both outcomes are exercised


def test_ray_triangle(orc, vec):
    rows = np.array([orc.ray_triangle(o, d, rec) for o, d, rec in zip(vec["tri_o"], vec["tri_d"], vec["tri_records"])])
    want = vec["tri_hit"]
    assert np.array_equal(rows[:, 0], want[:, 0])
    hit = want[:, 0] == 1
    assert 20 < hit.sum() < len(hit) - 20
    assert bits_equal(rows[hit], want[hit]), first_diff(rows[hit], want[hit])
    # a miss reports Hit(false, 0, 0, INF, materialIndex) in the shader; t and the material must agree
    assert bits_equal(rows[~hit][:, [1, 8]], want[~hit][:, [1, 8]])


def test_scene_traversal(orc, demo, vec):
    sc = pc.oracle_scene(orc, demo)
    rows = np.array([orc.ray_scene(sc, o, d)[0] for o, d in zip(vec["scene_o"], vec["scene_d"])])
    want = vec["scene_hit"]
    assert np.array_equal(rows[:, 0], want[:, 0]) and 30 < want[:, 0].sum() < len(want) - 10
    hit = want[:, 0] == 1
    assert bits_equal(rows[hit], want[hit]), first_diff(rows[hit], want[hit])


def test_stack_abort_and_tie_rule(orc, vec):
    """The 64-entry abort (raytrace.wgsl:167-171: best-so-far is returned when the stack holds 64
    entries at the top of the loop) on chain trees of depth 10..70, and the first-visited rule for
    equal t, as the executed shader decides them."""
    mats = layout.pack_materials([scenes.WHITE] * 4)
    aborted = 0
    for depth in (10, 62, 63, 64, 70):
        tris = np.frombuffer(vec[f"chain{depth}_tris"].tobytes(), layout.TRIANGLE)
        nodes = np.frombuffer(vec[f"chain{depth}_nodes"].tobytes(), layout.BVH_NODE)
        sc = orc.OracleScene(tris, mats, nodes)
        for row in vec[f"chain{depth}_hits"]:
            got, cnt = orc.ray_scene(sc, (row[0], row[1], 5.0), (0.0, 0.0, -1.0))
            assert bits_equal(got, row[2:]), (depth, row[:2], got, row[2:])
            aborted += cnt["stack_overflows"]
    assert aborted > 0                                        # the deep chains really hit the abort
    for order in (0, 1):
        tris = np.frombuffer(vec[f"tie{order}_tris"].tobytes(), layout.TRIANGLE)
        nodes = np.frombuffer(vec[f"tie{order}_nodes"].tobytes(), layout.BVH_NODE)
        got, _ = orc.ray_scene(orc.OracleScene(tris, mats, nodes), (0.0, 0.0, 2.0), (0.0, 0.0, -1.0))
        assert bits_equal(got, vec[f"tie{order}_hit"]) and got[8] == (2 - order)      # the RIGHT leaf is visited first


def test_camera_rays(orc, vec):
    for block, row in zip(vec["camera_uniforms"], vec["camera_uv_ray"]):
        got = orc.camera_ray(block.tobytes(), float(row[0]), float(row[1]))
        assert bits_equal(got, row[2:]), (row[:2], got, row[2:])


def test_environment_lookup(orc, demo, env, vec):
    sc = pc.oracle_scene(orc, demo, env)
    for block, row in zip(vec["env_uniforms"], vec["env_dir_uv_rgb"]):
        uv = orc.env_uv(block.tobytes(), row[:3])
        assert bits_equal(uv, row[3:5]), (row[:3], uv, row[3:5])
        rgb = orc.sample_env(sc, float(uv[0]), float(uv[1]))
        assert bits_equal(rgb, row[5:8]), (row[:3], rgb, row[5:8])


def _frame_ids(vec):
    return sorted(int(k[5:-9]) for k in vec.files if k.startswith("frame") and k.endswith("_uniforms"))


def test_whole_frames_match_the_executed_shader(orc, demo, env, vec):
    """computeMain run for every pixel by the interpreter == orc_raytrace (thin lens, 2 spp,
    env rotation / intensity, maxBounces 0, ragged sizes)."""
    sc = pc.oracle_scene(orc, demo, env)
    ids = _frame_ids(vec)
    assert len(ids) >= 6                                    # incl. the scalingFactor 0.7 sub-rectangle
    for i in ids:
        want = vec[f"frame{i}_image"]
        h, w = want.shape[:2]
        got, _ = orc.raytrace(sc, vec[f"frame{i}_uniforms"].tobytes(), w, h)
        assert bits_equal(got, want), f"frame {i}: " + first_diff(got, want)


def test_metal_and_emissive_materials(orc, demo, env, vec):
    """reflect(), the un-normalised mix() direction, specularColor and emission (raytrace.wgsl:380-395)."""
    mats = np.frombuffer(vec["metal_materials"].tobytes(), layout.MATERIAL)
    sc = orc.OracleScene(demo.triangles, mats, demo.nodes, env)
    want = vec["metal_image"]
    h, w = want.shape[:2]
    got, cnt = orc.raytrace(sc, vec["metal_uniforms"].tobytes(), w, h)
    assert bits_equal(got, want), first_diff(got, want)
    assert cnt["hits"] > 200 and want[..., :3].max() > 2.0             # emission reached the image


def test_dormant_environment_importance_sampling(orc, demo, env, vec):
    """raytrace.wgsl:315-367 with the commented-out call sites (:398, :402-404) enabled -- the
    generator removes the three comment markers from the in-memory shader text.  The oracle's
    optional mode (and the device's, mi3pt_set_env_sampling) reproduces it bit for bit."""
    cdf = capi.host_env_cdf(env)
    sc = orc.OracleScene(demo.triangles, demo.material_bytes, demo.nodes, env, cdf=cdf, env_sampling=True)
    want = vec["envsample_image"]
    h, w = want.shape[:2]
    got, cnt = orc.raytrace(sc, vec["envsample_uniforms"].tobytes(), w, h)
    assert bits_equal(got, want), first_diff(got, want)
    plain, _ = orc.raytrace(pc.oracle_scene(orc, demo, env), vec["envsample_uniforms"].tobytes(), w, h)
    assert not bits_equal(plain, want) and cnt["misses"] > 50          # the mode really changes the image


# ---------------------------------------------------------------- accumulate.wgsl

def test_accumulate_pass(orc, vec):
    cur, prev = vec["acc_cur"], vec["acc_prev"]
    h, w = cur.shape[:2]
    for (frame, enabled), want in zip(vec["acc_cases"], vec["acc_out"]):
        u = layout.UniformBlock(layout.ACCUMULATE_UNIFORMS).set({"resolution": [w - 1, h - 1], "frame": int(frame), "enabled": int(enabled)})
        got = orc.accumulate(u.tobytes(), w, h, cur, prev)
        inside = ~np.isnan(want[..., 0])
        assert inside.sum() == (w - 1) * (h - 1)                   # the bounds check left the last row / column alone
        assert bits_equal(got[inside], want[inside]), (frame, enabled, first_diff(got[inside], want[inside]))
        assert bits_equal(got[~inside], prev[~inside])             # the oracle keeps the previous value there


# ---------------------------------------------------------------- fullscreen.wgsl

def test_fullscreen_pass(orc, vec):
    tex = vec["fs_input"]
    h, w = tex.shape[:2]
    for (denoise, tonemap, scaling), want in zip(vec["fs_cases"], vec["fs_out"]):
        u = layout.UniformBlock(layout.FULLSCREEN_UNIFORMS).set(
            {"resolution": [w, h], "aspect": w / h, "scalingFactor": float(scaling), "denoise": int(denoise), "tonemapping": int(tonemap)})
        got, _ = orc.fullscreen(u.tobytes(), tex)
        assert bits_equal(got, want), f"denoise {denoise} tonemap {tonemap} scaling {scaling}: " + first_diff(got, want)


# ---------------------------------------------------------------- provenance

@pytest.mark.skipif(not os.path.isdir(REF_SHADERS), reason="the reference checkout is not on this machine")
def test_vectors_regenerate_from_the_reference_shaders(tmp_path, built, vec):
    """Re-runs the generator on the reference's shader files and compares with the committed vectors."""
    import subprocess
    import sys
    out = tmp_path / "v.npz"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_wgsl_vectors.py"), "/root/reference", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    fresh = np.load(out)
    assert sorted(fresh.files) == sorted(vec.files)
    for k in vec.files:
        a, b = fresh[k], vec[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert (bits_equal(a, b) if a.dtype == np.float32 else np.array_equal(a, b)), k


# ---------------------------------------------------------------- the device against the same vectors

@pytest.mark.gpu
def test_device_frames_match_the_executed_shader(gpu_ctx, demo, env, vec):
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    for i in _frame_ids(vec):
        want = vec[f"frame{i}_image"]
        h, w = want.shape[:2]
        ctx.resize(w, h)
        ctx.set_uniforms(capi.PASS_RAYTRACE, vec[f"frame{i}_uniforms"].tobytes())
        ctx.submit(capi.SUBMIT_RAYTRACE)
        got = ctx.read_texture(capi.TEX_OUTPUT)
        assert bits_equal(got, want), f"frame {i}: " + first_diff(got, want)
    ctx.upload_materials(np.frombuffer(vec["metal_materials"].tobytes(), layout.MATERIAL))
    want = vec["metal_image"]
    h, w = want.shape[:2]
    ctx.resize(w, h)
    ctx.set_uniforms(capi.PASS_RAYTRACE, vec["metal_uniforms"].tobytes())
    ctx.submit(capi.SUBMIT_RAYTRACE)
    got = ctx.read_texture(capi.TEX_OUTPUT)
    assert bits_equal(got, want), "metal / emissive frame: " + first_diff(got, want)
    # the dormant environment importance sampling, enabled on the device
    ctx.upload_materials(demo.material_bytes)
    ctx.upload_environment_cdf(capi.host_env_cdf(env))
    ctx.set_env_sampling(True)
    want = vec["envsample_image"]
    h, w = want.shape[:2]
    ctx.resize(w, h)
    ctx.set_uniforms(capi.PASS_RAYTRACE, vec["envsample_uniforms"].tobytes())
    ctx.submit(capi.SUBMIT_RAYTRACE)
    got = ctx.read_texture(capi.TEX_OUTPUT)
    ctx.set_env_sampling(False)
    assert bits_equal(got, want), "importance-sampling frame: " + first_diff(got, want)
    ctx.submit(capi.SUBMIT_RAYTRACE)
    assert not bits_equal(ctx.read_texture(capi.TEX_OUTPUT), want)      # and off again
    tex = vec["fs_input"]
    h, w = tex.shape[:2]
    ctx.resize(w, h)
    for (denoise, tonemap, scaling), want in zip(vec["fs_cases"], vec["fs_out"]):
        u = layout.UniformBlock(layout.FULLSCREEN_UNIFORMS).set(
            {"resolution": [w, h], "aspect": w / h, "scalingFactor": float(scaling), "denoise": int(denoise), "tonemapping": int(tonemap)})
        ctx.write_texture(capi.TEX_ACCUMULATION, tex)
        ctx.set_uniforms(capi.PASS_FULLSCREEN, u.tobytes())
        ctx.submit(capi.SUBMIT_FULLSCREEN)
        got = ctx.read_texture(capi.TEX_CANVAS)
        assert bits_equal(got, want), f"denoise {denoise} tonemap {tonemap} scaling {scaling}: " + first_diff(got, want)
    ctx.resize(64, 64)
