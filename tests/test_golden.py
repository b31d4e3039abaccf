"""Golden fixtures (tests/golden/demo_frames.npz, made by tests/golden/make_golden.py).

CPU: the oracle must still reproduce them (pins the oracle and the scene generator).
GPU: the HIP path must reproduce them bit for bit WITHOUT the oracle in the loop."""
import os

import numpy as np
import pytest

import ptcommon as pc
from mi3pt_host import capi

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_frames.npz"))
CASES = ["64_b1", "64_b4", "64_b8", "64_b4_dof", "256_b4"]
COUNTERS = ("rays", "box_tests", "tri_tests", "hits", "misses", "stack_overflows", "pixels")


def test_scene_generator_fingerprint(demo):
    assert np.frombuffer(demo.nodes.tobytes(), np.uint32).sum(dtype=np.uint64) == GOLD["scene_nodes_crc"][0]
    assert np.frombuffer(demo.triangles.tobytes(), np.uint32).sum(dtype=np.uint64) == GOLD["scene_tris_crc"][0]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(orc, demo, env, name):
    img = GOLD[name + "_image"]
    h, w = img.shape[:2]
    got, cnt = orc.raytrace(pc.oracle_scene(orc, demo, env), GOLD[name + "_uniforms"].tobytes(), w, h)
    assert pc.same_bits(got[..., :3], img), pc.describe_diff(got[..., :3], img)
    assert [cnt[k] for k in COUNTERS] == [int(v) for v in GOLD[name + "_counters"]]


def test_oracle_reproduces_golden_canvas(orc, demo, env):
    w = h = 64
    osc = pc.oracle_scene(orc, demo, env)
    acc = np.zeros((h, w, 4), np.float32)
    for frame in (2, 3, 4):
        img, _ = orc.raytrace(osc, pc.rt_uniforms(demo, w, h, frame=frame, bounces=4).tobytes(), w, h)
        acc = orc.accumulate(pc.acc_uniforms(w, h, frame).tobytes(), w, h, img, acc)
    assert pc.same_bits(acc[..., :3], GOLD["64_acc3_image"])
    _, c8 = orc.fullscreen(pc.fs_uniforms(w, h, 1.0, 1, 1).tobytes(), acc)
    assert np.array_equal(c8, GOLD["64_acc3_canvas_rgba8"])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 7], ids=["shipped", "reference-counter walk"])
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_golden(gpu_ctx, demo, env, name, variant):
    img = GOLD[name + "_image"]
    h, w = img.shape[:2]
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_kernel_variant(variant)
    ctx.set_storage(capi.STORAGE_F32)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    ctx.reset_counters()
    ctx.set_uniforms(capi.PASS_RAYTRACE, GOLD[name + "_uniforms"].tobytes())
    ctx.submit(capi.SUBMIT_RAYTRACE)
    got = ctx.read_texture(capi.TEX_OUTPUT)
    assert pc.same_bits(got[..., :3], img), pc.describe_diff(got[..., :3], img)
    assert (got[..., 3] == 1).all()
    cnt = ctx.counters()
    ctx.set_kernel_variant(0)
    # variant 7 runs exactly the reference's box / triangle tests; the shipped default also culls by distance
    pc.check_counters(cnt, dict(zip(COUNTERS, (int(v) for v in GOLD[name + "_counters"]))), culled=variant == 0)


@pytest.mark.gpu
def test_gpu_reproduces_golden_canvas_through_the_renderer(built, demo, env):
    """The whole Renderer.render() loop (3 frames) + fullscreen pass against the fixture."""
    from mi3pt_host import RaytracingCamera, RaytracingScene, Renderer
    r = Renderer.create()
    r.frames = 3
    r.scalingFactor = 1
    r.setUniforms("raytrace", {"maxBounces": 4, "envMapIntensity": 1.0})
    r.setUniforms("accumulate", {"enabled": 1})
    r.setUniforms("fullscreen", {"denoise": 1, "tonemapping": 1})
    r.resize(64, 64)
    scene = RaytracingScene(demo, env)
    scene.needsUpdate = True
    cam = RaytracingCamera(45.0)
    done = []
    r.on("complete", lambda: done.append(True))
    for _ in range(4):
        r.render(scene, cam)
    assert done == [True] and r.status == "idle"
    assert pc.same_bits(r.readAccumulation()[..., :3], GOLD["64_acc3_image"])
    assert np.array_equal(r.readCanvas(), GOLD["64_acc3_canvas_rgba8"])
    r.destroy()
