"""The driver's launch pattern (80 warm-up frames, then 256 + 64) with the five- / six-wave choice forced or automatic: run under
`rocprofv3 --kernel-trace` to see whether the second timed launch waits at the gate for the first one's drain (profiles/gate_trace.sh)."""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptcommon as pc
from mi3pt_host import capi, scenes
six = int(sys.argv[1]) if len(sys.argv) > 1 else -1
sc = scenes.dragon_class_scene(); sc.build_bvh(); env = scenes.synthetic_env()
W, H = 1920, 1080
ctx = capi.Context(0)
ctx.set_option(capi.OPT_SIX_WAVES, six)
pc.upload_scene(ctx, sc, env); ctx.resize(W, H)
def frames(f0, n):
    ctx.set_uniforms(capi.PASS_RAYTRACE, pc.rt_uniforms(sc, W, H, frame=f0, bounces=8).tobytes())
    ctx.set_uniforms(capi.PASS_ACCUMULATE, pc.acc_uniforms(W, H, f0).tobytes())
    ctx.submit_frames(3, n); ctx.flush()
frames(2, 80); ctx.sync()
frames(82, 256); frames(338, 64); ctx.sync()
print("gate", ctx.get_option(capi.OPT_GATE), "releases", ctx.get_option(capi.OPT_GATE_RELEASES), flush=True)
ctx.close()
