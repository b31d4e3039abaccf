"""The Node host (webgpu-pathtracer_amd/js): N-API addon + JS mirror of Renderer / Pass /
Scene.  CPU part: host logic and scene flattening against the Python generator.  GPU part:
the reference's render loop under Node, compared with the golden fixture."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import layout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "webgpu-pathtracer_amd", "js")
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def _node(args, **kw):
    return subprocess.run([NODE] + args, capture_output=True, text=True, timeout=300, **kw)


def test_addon_is_built(built):
    assert os.path.exists(os.path.join(JS, "mi3pt.node"))


def test_host_logic_under_node(built):
    r = _node([os.path.join(JS, "test", "host_cpu.test.js")])
    assert r.returncode == 0 and "host_cpu.test.js ok" in r.stdout, r.stdout + r.stderr


def test_js_scene_flattening_matches_python_generator(built, demo, tmp_path):
    """src/main.ts:36-75 built from PlaneGeometry / BoxGeometry / SphereGeometry in JS and
    flattened by RaytracePass (raytrace.ts:406-502) gives byte-identical buffers to the
    Python generator used by the parity tests; the BVH (native builder) too."""
    r = _node([os.path.join(JS, "tools", "dump_demo_scene.js"), str(tmp_path)])
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == {"triangles": 1998, "materials": 2, "nodes": 3995}
    assert (tmp_path / "triangles.bin").read_bytes() == demo.triangles.tobytes()
    assert (tmp_path / "materials.bin").read_bytes() == demo.material_bytes.tobytes()
    assert (tmp_path / "nodes.bin").read_bytes() == demo.nodes.tobytes()
    cam = np.frombuffer((tmp_path / "camera.bin").read_bytes(), layout.RAYTRACE_UNIFORMS)[0]
    assert tuple(cam["camera.position"]) == (0.0, 1.0, 4.0) and cam["camera.fov"] == 45.0
    assert np.array_equal(cam["camera.direction"], np.float32(demo.camera_direction()))   # (-0.0 == 0.0)


@pytest.mark.gpu
def test_render_loop_under_node_matches_golden(built, env, tmp_path):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "demo_frames.npz"))
    env_path = tmp_path / "env.f32"
    env_path.write_bytes(env.tobytes())
    out = str(tmp_path / "demo")
    r = _node([os.path.join(JS, "tools", "render_demo.js"), "--env", str(env_path), "--width", "64", "--height", "64",
               "--frames", "3", "--bounces", "4", "--out", out])
    assert r.returncode == 0, r.stdout + r.stderr
    summary = json.loads(r.stdout.strip().splitlines()[-1])
    assert summary["status"] == "idle" and summary["frame"] == 4
    assert summary["events"].count("complete") == 1 and summary["events"].count("progress") == 3
    assert summary["stats"] == {"Triangles": 1998, "Materials": 2, "BVH Nodes": 3995}
    assert summary["counters"]["pixels"] == 3 * 64 * 64
    acc = np.frombuffer(open(out + ".acc.f32", "rb").read(), np.float32).reshape(64, 64, 4)
    assert pc.same_bits(acc[..., :3], gold["64_acc3_image"]), pc.describe_diff(acc[..., :3], gold["64_acc3_image"])
    canvas = np.frombuffer(open(out + ".canvas.rgba8", "rb").read(), np.uint8).reshape(64, 64, 4)
    assert np.array_equal(canvas, gold["64_acc3_canvas_rgba8"])


@pytest.mark.gpu
def test_render_loop_under_node_on_a_device_group(built, env, tmp_path):
    """Renderer.create({ devices: [0, 0, 0] }): the reference-shaped loop (renderer.ts:366-395) drives a device group
    from Node -- three member contexts (all on the one GPU of the box), tiles dealt in 8-row blocks, one gather when
    the images are read -- and ends with the golden image and canvas, like the single-device loop."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "demo_frames.npz"))
    env_path = tmp_path / "env.f32"
    env_path.write_bytes(env.tobytes())
    out = str(tmp_path / "demo")
    r = _node([os.path.join(JS, "tools", "render_demo.js"), "--env", str(env_path), "--width", "64", "--height", "64",
               "--frames", "3", "--bounces", "4", "--out", out, "--devices", "0,0,0"])
    assert r.returncode == 0, r.stdout + r.stderr
    summary = json.loads(r.stdout.strip().splitlines()[-1])
    assert summary["status"] == "idle" and summary["frame"] == 4 and summary["counters"]["pixels"] == 3 * 64 * 64
    acc = np.frombuffer(open(out + ".acc.f32", "rb").read(), np.float32).reshape(64, 64, 4)
    assert pc.same_bits(acc[..., :3], gold["64_acc3_image"]), pc.describe_diff(acc[..., :3], gold["64_acc3_image"])
    canvas = np.frombuffer(open(out + ".canvas.rgba8", "rb").read(), np.uint8).reshape(64, 64, 4)
    assert np.array_equal(canvas, gold["64_acc3_canvas_rgba8"])


@pytest.mark.gpu
def test_render_loop_throughput_under_node(built, env, tmp_path):
    """The drop-in loop at speed: Renderer.render() driven like src/renderer.ts:366-395 / src/main.ts:387-400
    at 1920x1080, 8 bounces.  With the headless default (the canvas drawn once per launched batch) and with
    the fullscreen pass off the loop keeps the batched rate of the C ABI; presenting this very frame on
    every call (the reference's canvas semantics) costs a launch per frame.  All legs end with the same
    accumulation image, and both presenting legs with the same canvas."""
    env_path = tmp_path / "env.f32"
    env_path.write_bytes(env.tobytes())
    r = _node([os.path.join(JS, "tools", "bench_render_loop.js"), "--env", str(env_path), "--frames", "64"])
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    legs = j["legs"]
    print(json.dumps(j))
    assert j["same_canvas"] and j["same_accumulation"]
    assert len({l["rays"] for l in legs.values()}) == 1
    assert legs["no_present"]["mrays_per_s"] >= 0.9 * legs["c_abi"]["mrays_per_s"]
    assert legs["present_latest"]["mrays_per_s"] >= 0.75 * legs["c_abi"]["mrays_per_s"]
    assert legs["present_exact"]["mrays_per_s"] < legs["present_latest"]["mrays_per_s"]
