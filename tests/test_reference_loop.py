"""The Renderer frame state machine (renderer.ts:283-468: resize / render / reset / start / pause,
the frame counter whose first sampled frame is 2, progress, events, which passes are encoded)
against a trace produced by EXECUTING the reference's own methods (tests/golden/run_reference_loop.js
cuts them out of src/renderer.ts, strips the TypeScript annotations and runs them on recording
stubs).  tests/golden/renderer_loop_trace.json is that trace; the Node host and the Python host
must reproduce it call for call."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "golden", "run_reference_loop.js")
TRACE = os.path.join(ROOT, "tests", "golden", "renderer_loop_trace.json")
NODE = shutil.which("node")

SCRIPT = [["resize", 64, 48], ["render"], ["render"], ["render"], ["pause"], ["render"], ["start"], ["render"],
          ["set", "frames", 6], ["render"], ["render"], ["render"], ["render"], ["render"], ["reset"], ["render"],
          ["set", "scalingFactor", 0.5], ["render"], ["resize", 64, 48], ["resize", 32, 16], ["render"], ["pause"],
          ["reset"], ["render"], ["start"], ["render"], ["set", "frames", 1], ["render"], ["start"], ["render"],
          ["reset"], ["render"], ["render"], ["pause"], ["pause"], ["set", "frames", 0], ["start"], ["render"]]


def _run_node(tmp_path, with_reference):
    script = tmp_path / "script.json"
    script.write_text(json.dumps(SCRIPT))
    args = [NODE, HARNESS, str(script)] + (["/root/reference"] if with_reference else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout)


def _committed():
    with open(TRACE) as f:
        data = json.load(f)
    assert data["script"] == SCRIPT, "regenerate tests/golden/renderer_loop_trace.json (the script changed)"
    return data["trace"]


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_node_renderer_follows_the_executed_reference(built, tmp_path):
    assert _run_node(tmp_path, False)["mine"] == _committed()


def test_python_renderer_follows_the_executed_reference():
    from mi3pt_host import renderer as R

    class Ctx:                       # records what reaches the C ABI
        def __init__(self):
            self.mask = 0

        def submit(self, mask):
            self.mask |= mask

        def __getattr__(self, name):
            return lambda *a, **k: None

    class Scene:
        needsUpdate = False
        environment = None

    class Camera:
        fov, focalDistance, aperture = 45.0, 1.0, 0.0

        def world_position(self):
            return (0.0, 0.0, 0.0)

        def world_direction(self):
            return (0.0, 0.0, -1.0)

    ctx = Ctx()
    r = R.Renderer(ctx)
    r.passes["raytrace"].updateScene = lambda scene, camera: None      # the scene compile is not under test here
    pushed = []
    for name in ("raytrace", "accumulate", "fullscreen"):
        def recording(value, name=name, original=r.passes[name].setUniforms):
            pushed.append([name, json.loads(json.dumps(value))])
            original(value)
        r.passes[name].setUniforms = recording
    events = []
    for ev in ("start", "pause", "reset", "progress", "complete", "resize"):
        r.on(ev, lambda *a, ev=ev: events.append([ev, a[0]] if a else [ev]))
    trace = []
    for op in SCRIPT:
        del events[:]
        del pushed[:]
        if op[0] == "resize":
            r.resize(op[1], op[2])
        elif op[0] == "render":
            r.render(Scene(), Camera())
        elif op[0] == "set":
            setattr(r, op[1], op[2])
        else:
            getattr(r, op[0])()
        encoded = [n for bit, n in ((1, "raytrace"), (2, "accumulate"), (4, "fullscreen")) if ctx.mask & bit]
        ctx.mask = 0
        trace.append({"op": op, "events": [list(e) for e in events], "status": r.status, "frame": r.frame, "progress": r.progress,
                      "hasFramesToSample": r.hasFramesToSample, "encoded": encoded, "uniforms": list(pushed),
                      "size": [r.width, r.height, r.scaledWidth, r.scaledHeight]})
    want = _committed()
    for got, exp in zip(trace, want):
        assert got == exp, (got, exp)
    assert len(trace) == len(want)


@pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference/src"), reason="needs the reference checkout and node")
def test_trace_regenerates_from_the_reference_source(built, tmp_path):
    out = _run_node(tmp_path, True)
    assert out["reference"] == _committed()
    assert out["mine"] == out["reference"]
