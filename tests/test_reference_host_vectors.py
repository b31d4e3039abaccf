"""The native host-side scene compile against vectors produced by EXECUTING the reference's
TypeScript (tests/golden/run_reference_host.js cuts buildBVH / buildBVHRecursive / flattenBVH out
of src/passes/raytrace.ts and updateEnvironmentTexture out of src/renderer.ts, strips the type
annotations and runs them under Node with a minimal THREE shim).  tests/golden/host_vectors.npz
holds the outputs; the inputs are re-created from the generator's formulas."""
import importlib.util
import os
import shutil
import subprocess
import sys
import zlib

import numpy as np
import pytest

from mi3pt_host import capi, layout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VECTORS = os.path.join(ROOT, "tests", "golden", "host_vectors.npz")
GENERATOR = os.path.join(ROOT, "tests", "golden", "make_host_vectors.py")
HAVE_REFERENCE = os.path.isdir("/root/reference/src/passes") and shutil.which("node") is not None


def _generator():
    spec = importlib.util.spec_from_file_location("make_host_vectors", GENERATOR)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def vec():
    return np.load(VECTORS)


def test_native_bvh_builder_equals_the_executed_reference_builder(built, vec):
    """mi3pt_host_build_bvh_f64 (prefix/suffix boxes, thread pool) vs the reference's O(n^2)
    recursive builder + breadth-first flattener run as written: the 48-byte node records are
    identical, ties and all."""
    gen = _generator()
    sets = gen.triangle_sets()
    assert set(sets) >= {"demo", "soup", "grid", "tall", "pair", "single", "triple"}
    for name, pos in sets.items():
        want = vec[f"bvh_{name}_nodes"].tobytes()
        for threads in (1, 4):
            got = capi.host_build_bvh_f64(np.ascontiguousarray(pos, np.float64), threads)
            assert got.dtype == layout.BVH_NODE and len(got) == 2 * len(pos) - 1
            assert got.tobytes() == want, f"{name} ({len(pos)} triangles, {threads} threads)"


def test_native_env_cdf_equals_the_executed_reference_code(built, vec):
    """mi3pt_host_env_cdf vs Renderer.updateEnvironmentTexture run as written (its O(W^2 H) prefix
    sums, Float32Array roundings and V8's Math.sin): the 8 MiB CDF texture is bit-identical."""
    gen = _generator()
    for name, env in gen.environments().items():
        cdf = capi.host_env_cdf(env)
        flat = np.ascontiguousarray(cdf, np.float32).reshape(-1)
        sample, want = flat[:: gen.CDF_SAMPLE_STEP], vec[f"cdf_{name}_sample"]
        assert np.array_equal(sample.view(np.uint32), want.view(np.uint32)), name
        assert zlib.crc32(flat.tobytes()) == int(vec[f"cdf_{name}_crc"][0]), name


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference checkout and node on this machine")
def test_host_vectors_regenerate_from_the_reference_sources(tmp_path, built, vec):
    out = tmp_path / "h.npz"
    r = subprocess.run([sys.executable, GENERATOR, "/root/reference", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    fresh = np.load(out)
    assert sorted(fresh.files) == sorted(vec.files)
    for k in vec.files:
        assert fresh[k].tobytes() == vec[k].tobytes(), k
