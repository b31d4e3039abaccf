import sys, time, os
sys.path.insert(0, "webgpu-pathtracer_amd/py"); sys.path.insert(0, "tests")
import ptcommon as pc
from mi3pt_host import capi, scenes
sc = scenes.dragon_class_scene(); sc.build_bvh(); env = scenes.synthetic_env()
W, H = 1920, 1080
ctx = capi.Context(0)
pc.upload_scene(ctx, sc, env); ctx.resize(W, H)
def run(n, f0):
    t = time.perf_counter()
    for f in range(n):
        ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f0 + f, bounces=8).tobytes())
        ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f0 + f).tobytes())
        ctx.submit(3); ctx.sync()
    return (time.perf_counter() - t) / n * 1e3
run(32, 2)
for rnd in range(3):
    for tmo in (2000, 0):
        ctx.set_option(capi.OPT_GATE_TIMEOUT_MS, tmo)
        print(f"gate time-out {tmo:5d} ms: {run(64, 100):.4f} ms per frame (sync per frame)", flush=True)
ctx.close()
