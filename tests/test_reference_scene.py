"""The scene compile of the Node host (RaytracePass.flattenScene / packScene + the native BVH
builder) against the reference's own RaytracePass.updateScene, updateTriangleBuffer,
updateMaterialBuffer, updateBVHBuffer, buildBVH, buildBVHRecursive and flattenBVH EXECUTED under
Node (tests/golden/run_reference_scene.js: method texts cut out of src/passes/raytrace.ts at run
time, TypeScript annotations stripped, running on this repository's three.js stand-ins and a
recording GPUQueue).  Needs the reference checkout, so it runs on the build machine only; what it
establishes there is carried to other machines by tests/test_node_host.py (Node == Python buffers)
and tests/test_reference_host_vectors.py (builder bytes)."""
import json
import os
import shutil
import subprocess

import pytest

from test_asset_loaders import make_glb, make_obj

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "golden", "run_reference_scene.js")
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference/src/passes"),
                                reason="needs the reference checkout and node on this machine")


def _run(*args):
    r = subprocess.run([NODE, HARNESS, "/root/reference"] + list(args), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def _all_equal(out):
    return all(out[k] for k in ("trianglesEqual", "materialsEqual", "nodesEqual", "cameraEqual", "needsUpdateCleared"))


def test_default_scene_compiles_to_the_same_bytes(built):
    out = _run("demo")
    assert (out["triangles"], out["nodes"], out["materials"]) == (1998, 3995, 2)
    assert _all_equal(out), out


def test_loaded_models_compile_to_the_same_bytes(built, tmp_path):
    """A node hierarchy (TRS, matrix with negative scale, multi-primitive mesh), a second material
    met later in the traversal, an invisible mesh and a mesh with a foreign material."""
    glb, _, _ = make_glb()
    (tmp_path / "model.glb").write_bytes(glb)
    (tmp_path / "model.obj").write_text(make_obj())
    out = _run("model", str(tmp_path / "model.glb"))
    assert out["materials"] == 2 and out["triangles"] > 100 and _all_equal(out), out
    out = _run("model", str(tmp_path / "model.obj"))
    assert out["materials"] == 1 and _all_equal(out), out
