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
HAVE_REFERENCE = os.path.isdir("/root/reference/src/passes")
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_scene_hashes.json")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")
needs_reference = pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference checkout on this machine")


def _run(*args, root="/root/reference"):
    r = subprocess.run([NODE, HARNESS, root] + list(args), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def _all_equal(out):
    return all(out[k] for k in ("trianglesEqual", "materialsEqual", "nodesEqual", "cameraEqual", "needsUpdateCleared"))


@needs_reference
def test_default_scene_compiles_to_the_same_bytes(built):
    out = _run("demo")
    assert (out["triangles"], out["nodes"], out["materials"]) == (1998, 3995, 2)
    assert _all_equal(out), out


@needs_reference
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


def _cases(tmp_path):
    glb, _, _ = make_glb()
    (tmp_path / "model.glb").write_bytes(glb)
    (tmp_path / "model.obj").write_text(make_obj())
    return {"demo": ("demo",), "glb": ("model", str(tmp_path / "model.glb")), "obj": ("model", str(tmp_path / "model.obj"))}


def _check_against_fixture(tmp_path):
    cases = _cases(tmp_path)
    if HAVE_REFERENCE:          # regenerate: the fixture is what the REFERENCE's methods wrote, never this repository's output
        fresh = {}
        for name, args in cases.items():
            out = _run(*args)
            assert _all_equal(out), (name, out)
            fresh[name] = dict(out["reference"], counts=[out["triangles"], out["nodes"], out["materials"]])
        # (round-3 advice: a differing fixture used to be rewritten silently.)  A change in what the reference's methods write --
        # or in the harness -- FAILS; the tracked file is only regenerated on request: MI3PT_REGENERATE_FIXTURES=1
        if os.environ.get("MI3PT_REGENERATE_FIXTURES") == "1" or not os.path.exists(FIXTURE):
            with open(FIXTURE, "w") as f:
                json.dump(fresh, f, indent=1, sort_keys=True)
        assert json.load(open(FIXTURE)) == fresh, \
            "tests/golden/reference_scene_hashes.json differs from what the reference checkout produces now (MI3PT_REGENERATE_FIXTURES=1 rewrites it)"
    want = json.load(open(FIXTURE))
    for name, args in cases.items():
        out = _run(*args, root="-")
        w = want[name]
        assert [out["triangles"], out["nodes"], out["materials"]] == w["counts"], name
        for k in ("triangles", "materials", "nodes"):
            assert out["mine"][k] == w[k], f"{name}: {k} buffer differs from what the reference's scene compile wrote"
        assert out["mine"]["camera"] == w["camera"], name


def test_scene_compile_matches_the_reference_fixture(built, tmp_path):
    _check_against_fixture(tmp_path)


@pytest.mark.gpu
def test_scene_compile_matches_the_reference_fixture_on_the_gpu_box(built, tmp_path):
    """The same check where the GPU tests run (no reference checkout there)."""
    _check_against_fixture(tmp_path)
