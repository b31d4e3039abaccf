#!/usr/bin/env python3
"""Longer relatives of tests/test_gpu_configs.py, run by hand on a GPU box (the suite itself has to fit the driver's limit): the
shipped kernel against the CPU ORACLE on WHOLE images of BASELINE.json's big configurations, more frames than the tests hold.

    python tests/tools/deep_configs_gpu.py [config5 [frames]] [config4 [frames]] [config3 [frames]] [closeup [frames]]       (default: all four)

  config5   the 10 M-triangle forest, 3840 x 2160, 8 bounces: the whole image, `frames` frames (default 3; a frame is most of a
            minute of oracle on a 16-thread share)
  config4   the dragon-class scene with the thin lens, 3840 x 2160: the whole image, `frames` frames (default 32)
  config3   the dragon-class scene, 1920 x 1080: the whole image, `frames` frames (default 512: twice the stated 256 spp)
  closeup   the same scene with bench.py's close-up camera (the model fills the frame: the view for which the context switches to
            its deep-walk build by itself -- the line says which walk threshold the last launch ran), `frames` frames (default 64)

Bits and counters are compared after EVERY frame of the running mean (a progress line each); the first difference stops the run."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ("tests", os.path.join("webgpu-pathtracer_amd", "py"), "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
import ptcommon as pc  # noqa: E402
import pt_oracle as orc  # noqa: E402
from mi3pt_host import capi, scenes  # noqa: E402

MASK = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
ENV = scenes.synthetic_env()


def run(name, sc, w, h, nframes, **kw):
    t0 = time.time()
    walk_mins = set()
    ctx = capi.Context(0)
    pc.upload_scene(ctx, sc, ENV)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    ctx.reset()
    ctx.reset_counters()
    osc = pc.oracle_scene(orc, sc, ENV)
    acc = np.zeros((h, w, 4), np.float32)
    total = {}
    for f in range(2, 2 + nframes):          # renderer.ts:369-377: frame = 2, 3, ...
        u = pc.rt_uniforms(sc, w, h, frame=f, bounces=8, **kw)
        pc.gpu_frame(ctx, u, pc.acc_uniforms(w, h, f), MASK)
        part, cnt = orc.raytrace(osc, u.tobytes(), w, h, 0, 1, 8)
        acc = orc.accumulate(pc.acc_uniforms(w, h, f).tobytes(), w, h, part, acc, 0, 1, 8)
        for k, v in cnt.items():
            total[k] = total.get(k, 0) + v
        # every 16th frame (and the last) the running mean is read back and compared; the counters every frame
        if (f - 1) % 16 == 0 or f == 1 + nframes or nframes <= 8:
            got = ctx.read_texture(capi.TEX_ACCUMULATION)
            c = ctx.counters()
            walk_mins.add(ctx.last_launch()["walk_min"])
            if not pc.same_bits(got, acc):
                print(f"{name}: DIFFERENCE after frame {f}: {pc.describe_diff(got, acc)}", flush=True)
                return False
            pc.check_counters(c, total, culled=True, what=f"{name}, frame {f}")
            print(f"{name}: {f - 1} frames, {total['rays']} rays, bit-identical to the oracle, counters equal "
                  f"(variant {ctx.active_variant()}, walk thresholds so far {sorted(walk_mins)}, {c['box_tests'] / c['rays']:.1f} box tests per ray against the oracle's {total['box_tests'] / total['rays']:.1f}), {time.time() - t0:.0f} s", flush=True)
        else:
            print(f"{name}: frame {f} traced on both sides, {time.time() - t0:.0f} s", flush=True)
    ctx.close()
    return True


def main():
    args = sys.argv[1:]
    want = {}
    i = 0
    while i < len(args):
        n = None
        if i + 1 < len(args) and args[i + 1].isdigit():
            n = int(args[i + 1])
        want[args[i]] = n
        i += 2 if n is not None else 1
    if not want:
        want = {"config5": None, "config4": None, "config3": None, "closeup": None}
    ok = True
    dragon = None
    if "config3" in want or "config4" in want or "closeup" in want:
        dragon = scenes.dragon_class_scene()
        dragon.build_bvh()
    if ok and "config3" in want:
        ok = run("config 3 (870 k triangles, 1920x1080)", dragon, 1920, 1080, want["config3"] or 512)
    if ok and "closeup" in want:
        near = copy.copy(dragon)
        near.camera = dict(dragon.camera, position=(0.55, 0.62, 1.15), target=(0.0, 0.5, 0.0))      # (bench.py's CLOSEUP_CAMERA)
        ok = run("close-up (870 k triangles, 1920x1080, the model fills the frame)", near, 1920, 1080, want["closeup"] or 64)
    if ok and "config4" in want:
        focal = float(np.linalg.norm(np.array(dragon.camera["position"]) - np.array([0.0, 0.5, 0.0])))
        ok = run("config 4 (870 k triangles, thin lens, 3840x2160)", dragon, 3840, 2160, want["config4"] or 32, aperture=0.03, focal=focal)
    if ok and "config5" in want:
        forest = scenes.forest_scene()
        forest.build_bvh()
        print(f"forest: {len(forest.triangles)} triangles", flush=True)
        ok = run("config 5 (10 M-triangle forest, 3840x2160)", forest, 3840, 2160, want["config5"] or 3)
    print("deep configs: no difference" if ok else "deep configs: DIFFERENCE", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
