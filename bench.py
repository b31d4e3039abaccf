#!/usr/bin/env python3
"""Benchmark of the per-pixel ray-trace path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dragon|demo] [--scaling strong|weak]

One STEP = one pass of the hot path over one batch of input: FRAMES_PER_STEP = 16 consecutive
Renderer.render() frames (raytrace + accumulate passes, 1 sample per pixel each, whole image, 8
bounces) = 16 spp -- the unit the persistent raytrace kernel is launched on (one launch covers
one batch of frames; bit-identical to launching frame by frame).  K steps = 16 K frames; the
timed region ends with a flush + device sync.  (Round 1 counted a single frame as a step: with
the driver's 20 steps that is a 9 ms job, most of it launch ramp and drain.)

Headline workload (default) = the scene BASELINE.json's target is quoted on: the dragon-class
~870k-triangle mesh + environment map at 1920x1080 (configs[2]); `--workload demo` is
configs[1] (default demo mesh, 1,998 triangles) and is also measured, shorter, as `also.demo` -- and
as `also.demo_presenting_every_frame` with the reference's whole render(): a de-noised, tone-mapped canvas per frame.

N > 1 (one process per GPU: launched by torch.distributed.run -- or by bench.py itself when `python bench.py --gpus N`
is run plain, self_launch() -- ): image tiles are dealt to the
ranks in 8-row blocks, the scene is replicated, there is NO per-frame communication, and the job
ends with one RCCL gather of the HDR accumulation buffers to rank 0 (inside the timed region).
Default `--scaling strong`: the metric is Mrays/s AT 1920x1080, so the one 1080p image is split
N ways.  `--scaling weak` grows the image with N instead (N=2: 1920x2160, N=4: 3840x2160,
N=8: 3840x4320) so that every GPU always renders 1920x1080 pixels per frame.

Prints ONE JSON line on rank 0: metric Mrays/s (rays = raySceneIntersect calls, counted exactly
by the kernel), plus
  roofline     : the dominant kernel (the persistent raytrace kernel) against the roof that binds it.
                 It is NOT HBM-bound on this workload (the 181 MB scene sits in the 256 MiB Infinity
                 Cache; the algorithmic byte rate SURVEY.md 8(d) defines exceeds the HBM peak): what
                 it runs out of is vector-ALU lane-operations -- `bound: "valu"`, `achieved` = VALU
                 lane-operations per second (SQ_INSTS_VALU x 64 x lane utilisation / kernel time),
                 `peak` = CUs x 4 SIMDs x 2.4 GHz / 2 x 64 lanes, `frac` = achieved / peak =
                 valu_issue_frac x lane_utilisation.  The memory side is reported beside it:
                 `traffic` = fabric-side bytes (L2 misses) per launch from rocprofv3 PMC passes of THIS
                 command line (FETCH_SIZE x 1 + WRITE_SIZE: the factor calibrated on this kernel's own
                 access shapes, profiles/r05_fetch_calibration.log), `hbm` = traffic / kernel time
                 against 8 TB/s and against what a pure random-64-byte-gather kernel reaches, `l2` = (TCC_HIT + TCC_MISS) x 128 B / kernel
                 time against the guide's 34.5 TB/s, `algorithmic_GBps` as SURVEY.md defines it.
                 Kernel time = the launches' exclusive share of the GPU clock (`kernel_ms_exclusive`:
                 consecutive launches overlap at their tails), from HIP event pairs on the streams
                 the launches ran on.
  forest       : the same measurement for BASELINE.json's config 5 scene (10 M triangles, 2.08 GB:
                 the only one larger than the Infinity Cache, i.e. the one where HBM can bind), at
                 1920x1080 so that it fits the default run.
  cpu_baseline : the CPU oracle timed on a bounded sample of the same workload (rank 0, N = 1),
                 with its thread scaling.
The oracle is only ever the baseline / checker here, never the thing measured as `value`.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "webgpu-pathtracer_amd", "py"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0     # same guide: what a streaming kernel reaches
SHADER_CLOCK_HZ = 2.4e9         # peak engine clock used for the VALU issue rate
BLOCK_ROWS = int(os.environ.get("MI3PT_BENCH_BLOCK_ROWS", "8"))     # rows per block of the tile split (experiment knob)
BOUNCES = 8
FRAMES_PER_STEP = 16            # one step = one batch = one launch of the persistent kernel (single GPU)
L2_PEAK_GBS = 34500.0           # same guide: L2 (8 x 4 MiB), aggregate
FETCH_FACTOR = 1.0              # REQUESTED bytes per (FETCH_SIZE x 1024) for this kernel's divergent 64-byte gathers: measured, profiles/r05_fetch_calibration.log
FETCH_FACTOR_UPPER = 2.0        # ... if every such request moved a whole 128-byte line over the fabric (the size class the requests are tallied in): the
                                # upper bound of the bytes MOVED, and the guide's factor for streaming reads (MI355X_MICROARCH.md, HBM section)
GATHER64_HBM_GBS = 1640.0       # same log: random 64-byte gathers (4 x 16 B per lane) out of a 4 GiB table, useful bytes
GATHER64_MALL_GBS = 3500.0      # ... out of a 128 MiB table (resident in the Infinity Cache)
KERNEL_NEEDLE = "k_raytrace_sm"
# counter groups of the PMC passes: one rocprofv3 run each (FETCH_SIZE and WRITE_SIZE do not fit one pass)
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES",
               "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"),
              ("TCC_HIT_sum", "TCC_MISS_sum"),
              # the L2's fabric-side read requests by the size class they are tallied in (round-5 advice: FETCH_SIZE x 1 is the REQUESTED
              # bytes of this kernel's 64-byte gathers; every one of those requests is tallied as a 128-byte one)
              ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_sum"))
# a second camera on the headline scene: the model fills the frame (the reference's camera, main.ts:83-92, sees mostly
# floor and sky around the unit-height model: 1.86 rays per pixel, half of them single sky / floor segments)
CLOSEUP_CAMERA = {"position": (0.55, 0.62, 1.15), "target": (0.0, 0.5, 0.0)}
FOREST_IMAGE = (1920, 1080)     # the forest leg of the default run (config 5's scene; its own image is 3840x2160 on 8 GPUs)
FOREST_STEPS, FOREST_WARMUP = 8, 4


def image_size(n_gpus, scaling):
    if scaling == "strong":
        return 1920, 1080
    a, b = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}.get(n_gpus, (1, n_gpus))
    return 1920 * a, 1080 * b


def host_threads(world):
    """Threads of the host-side scene compile (the BVH builder's pool) for one rank: the processors this process may run on,
    at most 64, shared out among the ranks of the node (round 3: every rank of eight asked for all of them at once)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    return max(1, min(cores, 64) // max(world, 1))


def build_scene(workload, nthreads=0):
    from mi3pt_host import scenes
    if workload == "demo":
        sc = scenes.demo_scene()
    elif workload == "forest":
        sc = scenes.forest_scene()
    else:
        sc = scenes.dragon_class_scene()
        if workload == "closeup":
            sc.camera = dict(sc.camera, **CLOSEUP_CAMERA)
    sc.build_bvh(nthreads)
    return sc, scenes.synthetic_env()


def workload_name(workload, sc):
    if workload == "demo":
        return "default demo mesh (1,998 triangles) + synthetic env map"
    if workload == "forest":
        return f"instanced forest ({len(sc.triangles)} triangles, flattened) + synthetic env map"
    return (f"dragon-class procedural mesh ({len(sc.triangles)} triangles) + synthetic env map"
            + (", close-up camera (the model fills the frame)" if workload == "closeup" else ""))


def algorithmic_bytes(c):
    """SURVEY.md 8(d) / BASELINE.md: bytes in the reference's own layouts, each touched record
    counted once: 48 B per box test, 112 B per triangle test, one 64-B material per hit, four
    16-B env texels per miss, 16+16 B accumulator read-modify-write.  The counters are the
    kernel's own: the shipped walk skips boxes behind the closest hit, so this is below what
    the reference's walk would touch for the same image."""
    return 48 * c["box_tests"] + 112 * c["tri_tests"] + 64 * c["hits"] + 64 * c["misses"] + 32 * c["pixels"]


def frames_per_launch(cap):
    """Frames one launch covers: whole steps, as many as the library's batch capacity holds (16 on
    one GPU = one step per launch; a rank of an N-way split renders 1/N of the image per frame and
    batches up to N steps per launch); below 16 only when memory is short."""
    if cap >= FRAMES_PER_STEP:
        return cap - cap % FRAMES_PER_STEP
    return max(cap, 1)


class Job:
    """Scene + context + the frame loop, shared by the timed run and the PMC child runs."""

    def __init__(self, workload, width, height, rank=0, world=1, device=0, variant=0, stream_ptr=None, group=None, nthreads=0, scene=None):
        """group = [d0, d1, ...]: one device group (mi3pt_create_group) instead of one context; rank / world then stay 0 / 1.
        scene = (scene, env) of another Job of the same workload: not built again."""
        from mi3pt_host import capi, layout
        self.capi, self.layout = capi, layout
        self.workload, self.width, self.height = workload, width, height
        self.sc, self.env = scene if scene is not None else build_scene(workload, nthreads)
        self.ctx = capi.Context(devices=group, block_rows=BLOCK_ROWS) if group else capi.Context(device)
        if stream_ptr is not None:
            self.ctx.set_stream(stream_ptr)
        self.ctx.set_kernel_variant(variant)
        self.ctx.enable_timing(True)            # one HIP event pair per batched raytrace launch
        self.ctx.upload_bvh(self.sc.nodes)
        self.ctx.upload_triangles(self.sc.triangles)
        self.ctx.upload_materials(self.sc.material_bytes)
        self.ctx.upload_environment(self.env)
        if not group:
            self.ctx.set_tile(rank, world, BLOCK_ROWS)
        self.ctx.resize(width, height)
        self.frame = 2

    def rt_uniforms(self, frame):
        sc = self.sc
        u = self.layout.UniformBlock(self.layout.RAYTRACE_UNIFORMS)
        u.set({"resolution": [self.width, self.height], "aspect": self.width / self.height, "frame": frame,
               "maxBounces": BOUNCES, "samplesPerFrame": 1,
               "camera": {"position": sc.camera["position"], "direction": sc.camera_direction(),
                          "fov": sc.camera["fov"], "focalDistance": sc.camera["focalDistance"],
                          "aperture": sc.camera["aperture"]},
               "envMapIntensity": 1.0, "envMapRotation": 0.0})
        return u.tobytes()

    def acc_uniforms(self, frame):
        u = self.layout.UniformBlock(self.layout.ACCUMULATE_UNIFORMS)
        u.set({"resolution": [self.width, self.height], "frame": frame, "enabled": 1})
        return u.tobytes()

    def frames(self, n, per_launch, sync_each=False, present=False, plan=None):
        """n consecutive frames, `per_launch` per launch: one mi3pt_submit_frames call per launch
        (= that many Renderer.render() calls with only the frame counter moving), launched at
        once, nothing waited for (sync_each: wait after every launch -- the counter passes).
        present: every frame also encodes the fullscreen pass (de-noise + ACES), with the reference's canvas semantics
        (MI3PT_PRESENT_EXACT, the C ABI's default): a canvas drawn from the mean up to and including each frame."""
        capi, ctx = self.capi, self.ctx
        mask = capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE
        if present:
            mask |= capi.SUBMIT_FULLSCREEN
            f = self.layout.UniformBlock(self.layout.FULLSCREEN_UNIFORMS)
            f.set({"resolution": [self.width, self.height], "aspect": self.width / self.height, "scalingFactor": 1.0,
                   "denoise": 1, "tonemapping": 1})
            ctx.set_uniforms(capi.PASS_FULLSCREEN, f.tobytes())
        for rt, acc, k in (plan if plan is not None else self.plan(n, per_launch)):
            ctx.set_uniforms(capi.PASS_RAYTRACE, rt)
            ctx.set_uniforms(capi.PASS_ACCUMULATE, acc)
            ctx.submit_frames(mask, k)
            ctx.sync() if sync_each else ctx.flush()

    def plan(self, n, per_launch):
        """The uniform blocks of n consecutive frames cut into launches of per_launch: [(raytrace bytes, accumulate bytes, frames)].
        Inputs of the job (the timed region starts with them in hand, like the scene in HBM); advances the frame counter."""
        out, done = [], 0
        # EQUAL launches (round-5 verdict: the driver's 320 frames used to be a 256- and a 64-frame launch, which run two different
        # builds of the kernel -- six and five waves per SIMD -- and every per-launch figure averaged the two): the fewest launches
        # the batch capacity allows, all of the same whole number of steps where the frame count divides that way
        launches = max(-(-n // max(per_launch, 1)), 1)
        steps_total = n // FRAMES_PER_STEP
        even = n % FRAMES_PER_STEP == 0 and steps_total % launches == 0
        while done < n:
            k = (n // launches) if even else min(per_launch, n - done)
            out.append((self.rt_uniforms(self.frame), self.acc_uniforms(self.frame), k))
            self.frame += k
            done += k
        return out


def cpu_baseline(job, budget_s=20.0):
    """The CPU oracle timed on 1/16-image shards (8-row blocks dealt round robin, like the GPU tile split) of the same
    frames, with 1, 4, 16, 64 threads and the box's CPU share (cgroup quota, `pt_oracle.default_threads`) in turn, each for its share of the budget; one OpenMP team
    per shard call (tens to hundreds of ms of work each; the team's threads are kept between calls).  `value` is the
    best leg, `cores` its thread count: on a box whose CPU share is smaller than its processor count (a container
    quota) more threads than the share only add contention.  kind = "port": the oracle is a restatement, not the
    reference."""
    import pt_oracle as orc
    sc = job.sc
    osc = orc.OracleScene(sc.triangles, sc.material_bytes, sc.nodes, job.env)
    procs = max(orc.num_procs(), 1)
    share = orc.default_threads()            # the cgroup CPU quota where one is set, else the affinity mask (capped)
    legs = sorted({t for t in (1, 4, 16, 64, share) if t <= procs})
    shards = 16
    per_leg = budget_s / len(legs)
    scaling, n = {}, 0
    best = (0.0, 1, 0, 0, 0.0)
    for threads in legs:
        orc.set_num_threads(threads)
        rays = pixels = calls = 0
        t0 = time.perf_counter()
        while True:
            frame = 2 + n // shards
            _, cnt = orc.raytrace(osc, job.rt_uniforms(frame), job.width, job.height, n % shards, shards, BLOCK_ROWS)
            rays += cnt["rays"]
            pixels += cnt["pixels"]
            n += 1
            calls += 1
            dt = time.perf_counter() - t0
            if dt >= per_leg:
                break
        rate = rays / dt / 1e6
        scaling[str(threads)] = round(rate, 4)
        if rate > best[0]:
            best = (rate, threads, rays, pixels, dt)
    orc.set_num_threads(share)
    rate, threads, rays, pixels, dt = best
    return {"value": round(rate, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "thread_scaling_Mrays_per_s": scaling, "processors_visible": procs,
            "cpu_quota": orc.cpu_quota(), "oracle_default_threads": share,
            "sample": f"1/{shards}-image shards (8-row blocks, round robin) of the same {job.width}x{job.height} {BOUNCES}-bounce "
                      f"frames of the {job.workload} workload, {per_leg:.1f} s per thread count ({', '.join(str(t) for t in legs)}); "
                      f"best leg: {threads} threads, {pixels} pixel jobs, {rays} rays in {dt:.1f} s "
                      "(the oracle runs the reference's walk, without distance culling)"}


# ------------------------------------------------------------------------------------------
# PMC passes: this same command line re-run under `rocprofv3 --pmc` as child processes
# ------------------------------------------------------------------------------------------

def under_profiler():
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith("ROCPROF") for k in os.environ)


def inner_pmc(args):
    """Child mode: the job alone (no torch, no baseline, no output), for a counter pass.  One launch
    at a time: with counters on, rocprofv3 serialises kernels in the order it intercepts them on
    the two internal queues, which need not be the order they were enqueued in -- and a launch
    that is held (hipStreamWaitValue32) until its predecessor announces its drain can then end
    up in front of that predecessor: a deadlock, seen as passes that never finish.  The library
    therefore switches its launch gate off when a profiler is attached (mi3pt_create), and the
    child waits for every launch; per-launch counters are what a serialised run measures anyway."""
    width, height = FOREST_IMAGE if args.workload == "forest" else image_size(1, args.scaling)
    if args.image:
        width, height = (int(v) for v in args.image.lower().split("x"))
    tile_rank, tile_world = (int(v) for v in args.tile.split("/")) if args.tile else (0, 1)
    job = Job(args.workload, width, height, tile_rank, tile_world, variant=args.variant)
    per_launch = frames_per_launch(job.ctx.batch_capacity())
    job.frames(args.warmup * FRAMES_PER_STEP, per_launch, sync_each=True)
    job.frames(args.steps * FRAMES_PER_STEP, per_launch, sync_each=True)
    job.ctx.close()


def collect_pmc(args, timed_launches, log, workload=None, steps=None, warmup=None, timeout=150):
    """Runs the PMC passes; returns {counter: mean per timed launch} (the last `timed_launches`
    dispatches of the raytrace kernel in each pass are the timed ones) or {} when unavailable."""
    workload = workload or args.workload
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or timed_launches <= 0:
        return {}
    out = {}
    base = tempfile.mkdtemp(prefix="mi3pt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, counters in enumerate(PMC_PASSES):
            d = os.path.join(base, f"pass{i}")
            cmd = [rocprof, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--inner-pmc", "--steps", str(steps), "--warmup", str(warmup),
                   "--workload", workload, "--scaling", args.scaling, "--variant", str(args.variant)]
            if args.image and workload == args.workload:
                cmd += ["--image", args.image]
            if args.tile and workload == args.workload:
                cmd += ["--tile", args.tile]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            except (subprocess.TimeoutExpired, OSError) as e:
                log.append(f"pmc pass {counters}: {type(e).__name__}")
                break                   # a pass that had to be killed: start nothing else on this GPU
            if r.returncode != 0:
                log.append(f"pmc pass {counters}: rc {r.returncode}: {r.stdout.decode(errors='replace')[-300:]}")
                continue
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if KERNEL_NEEDLE in row["Kernel_Name"]:
                            per.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            for name, rows in per.items():
                rows.sort()
                vals = [v for _, v in rows[-timed_launches:]]
                if vals:
                    out[name] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return out


def traffic_from_file(key):
    """Fallback when no live PMC pass is possible (under a profiler, N > 1, rocprofv3 missing): the committed measurement
    for exactly this launch shape; for N > 1 -- where a rank cannot run counter passes of its own inside the job -- the
    entry of the same split measured rank by rank on ONE GPU (`bench.py --tile R/N`, profiles/pmc_rank_shapes.sh: the ranks
    of a tile split share nothing but the final gather, so a rank's kernel does the same work alone), with the nearest
    launch depth, its counters scaled to this job's frames per launch (they are proportional to the frames rendered)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            entries = json.load(f).get("entries", [])
    except (OSError, ValueError):
        return None
    for e in entries:
        if all(e.get(k) == v for k, v in key.items()):
            return e
    if key.get("n_gpus", 1) > 1:
        rest = {k: v for k, v in key.items() if k != "frames_per_launch"}
        near = [e for e in entries if all(e.get(k) == v for k, v in rest.items()) and e.get("frames_per_launch")]
        if near and key.get("frames_per_launch"):
            e = min(near, key=lambda e: abs(e["frames_per_launch"] - key["frames_per_launch"]))
            f = key["frames_per_launch"] / e["frames_per_launch"]
            unscaled = ("SQ_WAVES",)
            return dict(e, counters={k: (v if k in unscaled else v * f) for k, v in e["counters"].items()},
                        source=e.get("source", "") + f"; counters of a {e['frames_per_launch']:g}-frame launch scaled x{f:.3f} to {key['frames_per_launch']:g} frames")
    return None


def roofline_block(m, pmc, source, num_cus):
    """m: measurements of the timed run (see measure()); pmc: counter means per timed launch.
    Rates are per launch over the launches' exclusive share of the GPU clock (kernel_ms_exclusive = span from the
    first launch's start to the last one's end / launches: consecutive launches overlap at their tails, so the
    sum of their own durations, kernel_ms, counts the overlap twice).  The timed launches are EQUAL (Job.plan) and run one
    instantiation of the kernel (`launches`), so a per-launch mean is a mean over like things.
    Three fractions, by three rules, at top level (`frac_by_rule`): the vector-ALU lane-operation fraction (the roof that binds:
    `frac`), the counter-based HBM-side fraction north_star names (`hbm_counter`: requested bytes; `hbm_counter_upper`: if every
    request moved a 128-byte line), and SURVEY.md 8(d)'s algorithmic bytes over the HBM peak (above 1 on a scene the caches
    serve: reported because the rule asks for it, not a roof)."""
    kernel_ms = m["kernel_ms"]
    t_ms = m["kernel_ms_exclusive"] if m["kernel_ms_exclusive"] > 0 else kernel_ms
    t_s = t_ms * 1e-3
    alg_per_launch = algorithmic_bytes(m["counters"]) / max(m["launches"], 1)
    alg_gbps = alg_per_launch / t_s / 1e9 if t_s > 0 else None
    traffic = traffic_upper = None
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        traffic = int((FETCH_FACTOR * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0)
        traffic_upper = int((FETCH_FACTOR_UPPER * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0)
    # the fabric-side read requests by the size class they are TALLIED in (a separate pass); bytes if each moved its class's size
    rdreq = None
    if "TCC_EA0_RDREQ_128B_sum" in pmc:
        n32, n64, n128 = pmc.get("TCC_EA0_RDREQ_32B_sum", 0.0), pmc.get("TCC_EA0_RDREQ_64B_sum", 0.0), pmc["TCC_EA0_RDREQ_128B_sum"]
        by_class = 32.0 * n32 + 64.0 * n64 + 128.0 * n128
        rdreq = {"requests_32B": int(n32), "requests_64B": int(n64), "requests_128B": int(n128), "requests": int(pmc.get("TCC_EA0_RDREQ_sum", n32 + n64 + n128)),
                 "read_bytes_by_size_class": int(by_class),
                 "read_GBps_by_size_class": round(by_class / t_s / 1e9, 1) if t_s > 0 else None}
        if "WRITE_SIZE" in pmc:
            traffic_upper = int(by_class + pmc["WRITE_SIZE"] * 1024.0)
    hbm = traffic / t_s / 1e9 if traffic is not None and t_s > 0 else None
    hbm_upper = traffic_upper / t_s / 1e9 if traffic_upper is not None and t_s > 0 else None
    lane_peak = num_cus * 4 * SHADER_CLOCK_HZ / 2.0 * 64.0          # VALU lane-operations per second, all SIMDs
    issue_frac = lane_util = lane_rate = None
    if "SQ_INSTS_VALU" in pmc and t_s > 0:
        issue_frac = pmc["SQ_INSTS_VALU"] / t_s / (lane_peak / 64.0)
        if pmc.get("SQ_ACTIVE_INST_VALU"):
            lane_util = pmc.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
            lane_rate = pmc["SQ_INSTS_VALU"] * 64.0 * lane_util / t_s
    waves = m.get("waves_per_simd")
    block = {
        # the roof that binds (see real_bound): vector-ALU lane-operations
        "bound": "valu", "achieved": round(lane_rate / 1e12, 3) if lane_rate is not None else None,
        "peak": round(lane_peak / 1e12, 3), "unit": "Tlane-op/s",
        "frac": round(lane_rate / lane_peak, 4) if lane_rate is not None else None,
        "frac_rule": "SQ_INSTS_VALU x 64 x lane_utilisation / kernel time / (CUs x 4 SIMDs x 2.4 GHz / 2 x 64 lanes) = valu_issue_frac x lane_utilisation",
        "frac_by_rule": {
            "valu": round(lane_rate / lane_peak, 4) if lane_rate is not None else None,
            "hbm_counter": round(hbm / HBM_PEAK_GBS, 4) if hbm is not None else None,
            "hbm_counter_upper": round(hbm_upper / HBM_PEAK_GBS, 4) if hbm_upper is not None else None,
            "survey_8d_algorithmic": round(alg_gbps / HBM_PEAK_GBS, 3) if alg_gbps is not None else None,
            "rules": {"valu": "what binds: VALU issue x lane utilisation (frac_rule)",
                      "hbm_counter": "north_star's: (1 x FETCH_SIZE + WRITE_SIZE) KB per launch / kernel time / 8 TB/s -- REQUESTED bytes of the kernel's "
                                     "64-byte gathers (calibrated), a lower bound of the bytes moved over the fabric",
                      "hbm_counter_upper": "the same requests priced at the size class they are tallied in (128 B each): an upper bound of the bytes moved",
                      "survey_8d_algorithmic": "SURVEY.md 8(d): bytes the walk touches in the reference's layouts / kernel time / 8 TB/s; > 1 where L1 / L2 / "
                                               "Infinity Cache serve them (a 181 MB scene): not a roof, reported because the rule defines it"}},
        "traffic": traffic, "traffic_source": source,
        "traffic_if_every_request_moves_its_size_class": traffic_upper,
        "traffic_rule": f"traffic = ({FETCH_FACTOR:g} x FETCH_SIZE + WRITE_SIZE) KB per launch, separate rocprofv3 --pmc passes, mean over the (equal) timed launches: the "
                        "bytes this kernel REQUESTS from the fabric.  FETCH_SIZE = 64 B x L2-to-fabric read requests; calibrated on known byte counts in "
                        "this kernel's own access shapes (profiles/fetch_calibration.hip, r05_fetch_calibration.log): requested bytes / FETCH_SIZE = 1.000 "
                        "for divergent gathers of 64-byte records (one request per record), 2.000 for wide coalesced streaming reads (MI355X_MICROARCH.md's "
                        "case), which this kernel does not make.  That is a LOWER bound of the bytes moved: the same log shows every one of those requests "
                        "tallied as a 128-byte request (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ, 64B ~ 0, 32B = 0: collected live here, `fabric_read_requests`), "
                        f"and no counter says whether a 128-byte-class request for a 64-byte record moves 64 or 128 bytes -- priced at its class the "
                        f"traffic is `traffic_if_every_request_moves_its_size_class` (= {FETCH_FACTOR_UPPER:g} x FETCH_SIZE + WRITE_SIZE), the UPPER bound; "
                        "`hbm.achieved` / `hbm.frac` use the lower, `hbm.upper_*` the upper.  Either way these are L2 MISSES, whoever serves them: the same "
                        "gathers out of a table resident in the Infinity Cache count the same (no TCC counter separates the two)",
        "fabric_read_requests": rdreq,
        "hbm": {"achieved": round(hbm, 1) if hbm is not None else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(hbm / HBM_PEAK_GBS, 4) if hbm is not None else None,
                "upper_achieved": round(hbm_upper, 1) if hbm_upper is not None else None,
                "upper_frac": round(hbm_upper / HBM_PEAK_GBS, 4) if hbm_upper is not None else None,
                "peak_achievable": HBM_ACHIEVABLE_GBS,
                "frac_of_achievable": round(hbm / HBM_ACHIEVABLE_GBS, 4) if hbm is not None else None,
                # what a kernel that does nothing but this access shape reaches (profiles/r05_fetch_calibration.log): random 64-byte
                # records, four 16-byte loads per lane, one-wave workgroups -- out of a 4 GiB table (HBM) and out of a 128 MiB one
                # (Infinity Cache), in REQUESTED bytes like `achieved`.  The roof of a gather-bound walk; the byte peak above is a streaming kernel's
                "random_64B_gather_roof_GBps": {"from_hbm": GATHER64_HBM_GBS, "from_infinity_cache": GATHER64_MALL_GBS},
                "frac_of_gather_roof": {"from_hbm": round(hbm / GATHER64_HBM_GBS, 4) if hbm is not None else None,
                                        "from_infinity_cache": round(hbm / GATHER64_MALL_GBS, 4) if hbm is not None else None}},
        "kernel": m["kernel"], "kernel_ms": round(kernel_ms, 4), "kernel_ms_exclusive": round(m["kernel_ms_exclusive"], 4),
        "launches_timed": m["launches"], "frames_per_launch": m["frames_per_launch"],
        # one block per timed launch: which instantiation it ran (the launches of a job are equal, so: the same one)
        "launches": [{"frames": m["frames_per_launch"], "instantiation": m["kernel"].split(" (")[0], "waves_per_simd": waves,
                      "workgroups": m.get("workgroups"), "kernel_ms": round(kernel_ms, 4)} for _ in range(int(m["launches"]))],
        "kernel_ms_all_launches": round(m["kernel_ms_all"], 4), "launches_all": m["launches_all"],
        # what SURVEY.md 8(d) calls the achieved figure: bytes the algorithm touches (in the reference's layouts, from the
        # kernel's own counters) per second.  Served by L1 / L2 / Infinity Cache: not comparable with an HBM peak.
        "algorithmic_bytes_per_launch": int(alg_per_launch),
        "algorithmic_GBps": round(alg_gbps, 1) if alg_gbps is not None else None,
        "traffic_over_algorithmic": round(traffic / alg_per_launch, 3) if traffic is not None and alg_per_launch > 0 else None,
        "bytes_per_ray": round(algorithmic_bytes(m["counters"]) / max(m["counters"]["rays"], 1), 1),
        "box_tests_per_ray": round(m["counters"]["box_tests"] / max(m["counters"]["rays"], 1), 2),
        "tri_tests_per_ray": round(m["counters"]["tri_tests"] / max(m["counters"]["rays"], 1), 2),
    }
    if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc and t_s > 0:
        req = pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]
        l2 = req * 128.0 / t_s / 1e9
        block["l2"] = {"achieved": round(l2, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(l2 / L2_PEAK_GBS, 4),
                       "hit_rate": round(pmc["TCC_HIT_sum"] / req, 4) if req > 0 else None,
                       "rule": "(TCC_HIT_sum + TCC_MISS_sum) x 128 B per launch / kernel time; hit rate = TCC_HIT / (TCC_HIT + TCC_MISS)"}
    if issue_frac is not None:
        block["valu_issue_frac"] = round(issue_frac, 4)
        block["valu_insts_per_launch"] = int(pmc["SQ_INSTS_VALU"])
        if lane_util is not None:
            block["lane_utilisation"] = round(lane_util, 4)
        if pmc.get("SQ_WAVE_CYCLES"):
            wc = pmc["SQ_WAVE_CYCLES"]
            block["wave_cycles"] = {"issuing": round(pmc.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
                                    "waiting_for_memory": round(pmc.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                                    "issue_stalled": round(pmc.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)}
        block["pmc_counters"] = {k: round(v, 1) for k, v in sorted(pmc.items())}
        block["real_bound"] = ("vector-ALU lane-operations, reached through per-wave latency: each step is a dependent chain of one LDS and "
                               "one L2 / fabric round trip (a divergent 64-B packet gather) and a few hundred instructions that a single "
                               f"wave issues at one per 4 cycles, with {waves if waves else 'five or six'} waves per SIMD "
                               f"({ {5: 96, 6: 80}.get(waves, '80 - 96') } VGPRs) to overlap them; frac = VALU issue x lane "
                               "utilisation is the fraction of the machine's lane-op rate in use; HBM-side traffic and L2 requests are "
                               "reported beside it (hbm, l2) and are not what the kernel runs out of on a scene that fits the Infinity Cache")
    return block


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it (round-5 verdict: that command died here).  This process
    becomes the launcher and never touches the GPU: it builds (compilers only), picks a free rendezvous port on 127.0.0.1 and
    starts N fresh children of this same command line, one rank per GPU, with the variables torch.distributed.run would
    set; rank 0's stdout (the one JSON line) is this process's stdout, the other ranks' stdout goes to stderr.  Returns
    the first non-zero exit code of a child (the other children are then ended by PID), 0 when all succeed.  No exec,
    no re-launch of a process that has initialised HIP."""
    import signal
    import socket
    import __graft_entry__ as ge
    ge.build(load=False)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MI3PT_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (this pool's hosts only support dmabuf IPC: RCCL between processes needs it)
        env.setdefault("OMP_NUM_THREADS", "1")                  # (what torch.distributed.run sets for its workers; the scene compile has its own thread pool)
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                         stdout=None if r == 0 else sys.stderr))
    rc, grace = 0, None
    try:
        while any(c.poll() is None for c in children):
            failed = [c.returncode for c in children if c.poll() is not None and c.returncode != 0]
            if failed and rc == 0:
                rc, grace = failed[0], time.monotonic() + 20.0      # the others usually fail at their next collective by themselves
            if grace is not None and time.monotonic() > grace:
                for c in children:
                    if c.poll() is None:
                        c.send_signal(signal.SIGTERM)
                grace = time.monotonic() + 10.0
                for c in children:
                    try:
                        c.wait(timeout=max(grace - time.monotonic(), 0.1))
                    except subprocess.TimeoutExpired:
                        c.kill()
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        for c in children:
            if c.poll() is None:
                c.kill()
        rc = 130
    for c in children:
        c.wait()
        if rc == 0 and c.returncode != 0:
            rc = c.returncode
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)      # x 16 frames = 256 spp (BASELINE.json configs[2])
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dragon", choices=["demo", "dragon", "closeup", "forest"])
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (children of this process)")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements (demo scene, close-up camera)")
    ap.add_argument("--no-gather-check", action="store_true", help="N > 1: skip the re-rendering of two ranks' shares on rank 0 after the timed region (it renders "
                                                                    "every frame of the job twice more, outside the timed region)")
    ap.add_argument("--no-forest", action="store_true", help="skip the forest leg (config 5's scene: ~1 min of scene build, render and counter passes)")
    ap.add_argument("--image", default=None, help="WxH override (experiments only)")
    ap.add_argument("--tile", default=None, help="R/N: render only rank R's share of an N-way tile split on this one GPU, no gather "
                                                 "(experiments only: profiles/scaling_model.py predicts the multi-GPU curve from it)")
    ap.add_argument("--group", action="store_true", help="ONE process drives all N GPUs through a device group (mi3pt_create_group: the library's own "
                                                         "tile split + peer-copy gather) instead of one process per GPU over torch.distributed; launch as "
                                                         "`python bench.py --gpus N --group` (MI3PT_BENCH_REHEARSAL=1: all members on device 0)")
    ap.add_argument("--inner-pmc", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.inner_pmc:
        inner_pmc(args)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    group_devices = None
    if args.group:
        if world != 1:
            raise SystemExit("--group is one process for all GPUs: do not launch it through torch.distributed.run")
        group_devices = [0] * args.gpus if os.environ.get("MI3PT_BENCH_REHEARSAL") == "1" else list(range(args.gpus))
    elif "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches the GPU)
        raise SystemExit(self_launch(args.gpus))
    elif world != args.gpus:
        args.gpus = world
    # ONE JSON line on stdout, whatever the libraries underneath print: from here on file descriptor 1 is stderr (gloo announces its
    # connections on stdout -- "[Gloo] Rank 0 is connected to ..." -- and a backend's C++ side does not ask Python), and the line is
    # written to the real stdout, kept aside, at the end.  (The launcher parent of a plain multi-GPU run has returned above: its ranks
    # each do this for themselves.)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("MI3PT_BENCH_ECHO_RANK") == "1":      # (tests/test_bench_launcher.py)
        print(f"bench.py rank {rank}/{world} local {local_rank} rendezvous {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
              file=sys.stderr, flush=True)
    # Build (make / g++ children) BEFORE anything touches the GPU; the other ranks meet rank 0 at
    # the process-group rendezvous below, i.e. after the build.  The library itself is loaded
    # after torch (two HIP runtimes in one process: the first one loaded has to be torch's).
    import __graft_entry__ as ge
    if rank == 0:
        ge.build(load=False)

    import torch
    import torch.distributed as dist
    from mi3pt_host import capi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # Rehearsal mode (one GPU box): MI3PT_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # uses gloo (host-staged gather) instead of RCCL, to exercise the N > 1 code path.
    rehearsal = os.environ.get("MI3PT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # MI3PT_BENCH_RCCL_SELFTEST=1 (one rank): the N > 1 code path on the REAL backend -- process group over RCCL, the gather of device
    # tensors on the context's stream, the reductions -- with a world of one (what a one-GPU box can run of it: two ranks on one GPU
    # are refused by RCCL, which is why the N = 2 rehearsal goes over gloo)
    selftest = world == 1 and not group_devices and os.environ.get("MI3PT_BENCH_RCCL_SELFTEST") == "1"
    use_dist = world > 1 or selftest
    if selftest:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            free_port = sk.getsockname()[1]
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", str(free_port))):
            os.environ.setdefault(k, v)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()
    capi.load_library()

    n_gpus = args.gpus if group_devices else world
    threads = host_threads(world)
    base_width, base_height = FOREST_IMAGE if args.workload == "forest" else image_size(n_gpus, args.scaling)
    if args.image:
        base_width, base_height = (int(v) for v in args.image.lower().split("x"))
    width, height = base_width, base_height
    red_dev = "cpu" if rehearsal else "cuda"

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(workload, steps, warmup, gather, size=None, present=False):
        """The timed job for one workload: `warmup` untimed steps, then EXACTLY `steps` steps of 16 frames
        (+ the one gather when N > 1) between barrier + synchronize on both sides."""
        width, height = size or (base_width, base_height)
        stream = torch.cuda.Stream()       # the context's main stream
        tile_rank, tile_world = (int(v) for v in args.tile.split("/")) if args.tile else (rank, world)
        use_group = group_devices if (gather and group_devices) else None
        job = Job(workload, width, height, tile_rank, tile_world, local_rank, args.variant, None if use_group else stream.cuda_stream,
                  group=use_group, nthreads=threads)
        ctx = job.ctx
        if present:
            ctx.enable_timing(False)       # (an event pair around each of the 2 x frames little passes costs this leg 10-15 %)
        accum = None
        if not use_group:
            # the accumulation image lives in a torch tensor so RCCL can gather it in place
            accum = torch.zeros((ctx.local_rows, width, 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()           # (the fill ran on torch's stream, the passes run on the context's)
            ctx.bind_accumulation(accum.data_ptr(), accum.numel() * 4)
        max_rows = max(capi.tile_local_rows(height, r, tile_world, BLOCK_ROWS) for r in range(tile_world))      # (the deal goes back and forth: rank 0 need not hold the most rows)
        send = gathered = None
        gathered_host = []                 # (rehearsal: the gloo gather's host tensors on rank 0)
        ev_g0, ev_g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if use_dist and gather:
            send = torch.zeros((max_rows, width, 4), dtype=torch.float32, device="cuda")
            if rank == 0:
                gathered = [torch.empty_like(send) for _ in range(world)]

        def exchange():
            # the one exchange of the job: HDR accumulation buffers -> rank 0 over xGMI
            with torch.cuda.stream(stream):
                ev_g0.record(stream)
                send[: ctx.local_rows].copy_(accum)
                if rehearsal:
                    stream.synchronize()
                    host = send.cpu()
                    if rank == 0:
                        gathered_host[:] = [torch.empty_like(host) for _ in range(world)]
                    dist.gather(host, gathered_host if rank == 0 else None, dst=0)
                else:
                    dist.gather(send, gathered, dst=0)
                ev_g1.record(stream)

        per_launch = frames_per_launch(ctx.batch_capacity())
        if send is not None and warmup > 0:
            # warm-up of the exchange: the first gather on a communicator sets up RCCL's point-to-point channels (tens of
            # ms), which is not part of a steady-state job.  BEFORE the warm-up steps, so that those end right in front of
            # the timed region: a GPU that has idled for some tens of ms runs its next 12 ms job ~1 ms slower
            # (profiles/r03_h_first_job.log)
            exchange()
            sync_all()
        job.frames(warmup * FRAMES_PER_STEP, per_launch, present=present)
        ctx.sync()
        ctx.reset_counters()
        warm_ms, warm_launches, _ = ctx.raytrace_launch_stats()      # (the warm-up's launches: for the all-launch average below)
        ctx.raytrace_launch_stats(reset=True)

        timed_plan = job.plan(steps * FRAMES_PER_STEP, per_launch)
        sync_all()
        t0 = time.perf_counter()
        job.frames(steps * FRAMES_PER_STEP, per_launch, present=present, plan=timed_plan)      # ends with a flush: everything is launched, nothing waited for
        t_submitted = time.perf_counter()
        if send is not None:
            exchange()
        elif use_group:
            ctx.accumulation_device_ptr()  # the group's one exchange: every member's rows -> the presenting context's image (peer copies), waited for
        sync_all()
        elapsed = time.perf_counter() - t0
        submit_ms = (t_submitted - t0) * 1e3
        gather_ms = ev_g0.elapsed_time(ev_g1) if send is not None else None

        counters = ctx.counters()
        launch_ms_total, launches, launch_frames = ctx.raytrace_launch_stats()
        span_ms = ctx.raytrace_launch_span()
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        c = torch.tensor([counters[k] for k in capi.COUNTER_NAMES], dtype=torch.float64, device=red_dev)
        # per rank: wall time of the timed region, host time until everything was launched, GPU-clock span of the rank's raytrace
        # launches, GPU time of its part of the gather (copy into the send buffer + the RCCL call on its stream)
        mine = torch.tensor([elapsed * 1e3, submit_ms, span_ms, gather_ms if gather_ms is not None else -1.0, float(ctx.local_rows)],
                            dtype=torch.float64, device=red_dev)
        per_rank = [mine]
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            per_rank = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(per_rank, mine)
        total = dict(zip(capi.COUNTER_NAMES, (int(x) for x in c.tolist())))
        ranks = [dict(zip(("elapsed_ms", "submit_ms", "launch_span_ms", "gather_ms", "rows"), (round(float(v), 4) for v in r.tolist()))) for r in per_rank]
        for r in ranks:
            r["rows"] = int(r["rows"])
            if r["gather_ms"] < 0:
                r["gather_ms"] = None
        m = {"job": job, "elapsed": float(t.item()), "total": total, "counters": counters, "per_rank": ranks, "host_threads": threads,
             "kernel_ms": launch_ms_total / max(launches, 1), "launches": int(launches),
             "kernel_ms_exclusive": span_ms / max(launches, 1),
             # average over EVERY launch of the process, warm-up included -- the population `rocprofv3 --stats` averages
             # (the warm-up's launches can be shorter: W steps need not be a whole number of launches)
             "kernel_ms_all": (launch_ms_total + warm_ms) / max(launches + warm_launches, 1), "launches_all": int(launches + warm_launches),
             "frames_per_launch": round(launch_frames / max(launches, 1), 2),
             "variant": ctx.active_variant()}
        # which instantiation the timed launches ran (mi3pt_debug_last_launch: the launches of a job are equal -- Job.plan -- so the last
        # one stands for all): template arguments DEFER, CULL, WIDE, FILT, YMAX, DIAG, TOPLDS, LITE, CW, WMIN, WAVES[, W8] -- csrc/pt_kernels.hip.
        # WAVES from the grid: a launch that fills the machine runs CUs x 4 SIMDs x WAVES one-wave workgroups
        last = ctx.last_launch() if not use_group else {"variant": ctx.active_variant(), "lean": None, "workgroups": None, "waves_per_simd": 0, "ymax": None, "walk_min": 0}
        v, wg = last["variant"], last["workgroups"]
        waves = last["waves_per_simd"] or None                   # (MI3PT_OPT_LAST_BUILD: the template arguments rocprofv3 prints for this launch)
        ymax = {True: "true", False: "false", None: "YMAX"}[last["ymax"]]
        wmin = last["walk_min"] or "WMIN"
        names = {14: f"k_raytrace_sm<true,true,true,true,{ymax},false,false,false,true,{wmin},{waves or 'WAVES'},true>",
                 13: f"k_raytrace_sm<true,true,true,true,{ymax},false,false,false,true,{wmin},{waves or 'WAVES'}>",
                 12: "k_raytrace_sm<true,true,true,true,true,false>", 11: "k_raytrace_sm<true,true,true,true,false,false>",
                 10: "k_raytrace_sm<true,true,true,false,false,false>", 9: "k_raytrace_sm<true,true,false,false,false,false>",
                 7: "k_raytrace_sm<true,false,false,false,false,false>", 4: "k_raytrace_sm<false,false,false,false,false,false>"}
        m["kernel"] = (names.get(v, f"raytrace kernel variant {v}") + " (persistent raytrace kernel: per-lane state machine, deferred-leaf walk"
                       + (" with exact-image distance culling" if v >= 9 else "") + (" on 4-ary wide packets" if 10 <= v <= 13 else "")
                       + (", filtered slab test" if v in (11, 12) else "") + (", one-axis culling condition" if v == 12 else "")
                       + (": COMPRESSED packets (64 B, boxes on an 8-bit grid rounded outward, conservative test; the exact test on the leaf's own box in the triangle step)" if v == 13 else "")
                       + (" on EIGHT-wide compressed packets (80 B; hit masks in octant order, one 64-bit node entry and one 32-bit leaf entry per step: the experiment of round 6, not the default)" if v == 14 else "")
                       + f"; lean build, {waves if waves else 'five or six'} waves per SIMD" + (" (80 registers: launches of >= 1.5 M jobs)" if waves == 6 else " (96 registers)" if waves == 5 else "")
                       + "; YMAX = the one-axis culling condition where the scene's margins allow it; batched frames)")
        m["lean"], m["waves_per_simd"], m["workgroups"] = last["lean"], waves, wg
        # ---- is the gathered image the right image?  (outside the timed region; round-4 verdict: the N > 1 line gathered and
        # discarded.)  Rank 0 renders the shares of two OTHER ranks again, alone, on its own GPU -- every row, every frame of the
        # job (warm-up included: the accumulation image is the mean since frame 2) -- and compares them bit for bit with what
        # arrived; its own rows against its own image.  A rank's pixels depend on nothing but (global pixel, frame).
        if gather and not args.tile and not args.no_gather_check and (send is not None or use_group):
            verdict = None
            if rank == 0:
                import numpy as np
                from mi3pt_host import tiles
                nframes = (warmup + steps) * FRAMES_PER_STEP
                if use_group:
                    whole = ctx.read_texture(capi.TEX_ACCUMULATION)
                    parts = {r: whole[tiles.local_rows_of(height, r, len(use_group), BLOCK_ROWS)] for r in range(len(use_group))}
                    n_split = len(use_group)
                else:
                    got = gathered_host if rehearsal else gathered
                    n_split = world
                    parts = {r: got[r].cpu().numpy()[: capi.tile_local_rows(height, r, world, BLOCK_ROWS)] for r in range(world)}
                    own = accum.cpu().numpy()
                    verdict = {"own_rows_identical": bool(np.array_equal(parts[0].view(np.uint32), own.view(np.uint32)))}
                checked = sorted(r for r in {1, n_split - 1} - {0} if r < n_split)
                same = True
                for r in checked:
                    cj = Job(workload, width, height, r, n_split, local_rank, args.variant, nthreads=threads, scene=(job.sc, job.env))
                    cj.frames(nframes, frames_per_launch(cj.ctx.batch_capacity()))
                    img = cj.ctx.read_texture(capi.TEX_ACCUMULATION)
                    cj.ctx.close()
                    same = same and img.shape == parts[r].shape and bool(np.array_equal(img.view(np.uint32), parts[r].view(np.uint32)))
                verdict = dict(verdict or {}, ranks_rerendered=checked, frames=nframes, rows=[int(parts[r].shape[0]) for r in checked],
                               identical=same)
                m["gather_verified"] = same and verdict.get("own_rows_identical", True)
                m["gather_check"] = verdict
        if accum is not None:
            ctx.bind_accumulation(None, 0)
        return m

    m = measure(args.workload, args.steps, args.warmup, gather=True)
    job = m["job"]
    out = None
    if rank == 0:
        rays, elapsed, steps = m["total"]["rays"], m["elapsed"], max(args.steps, 1)
        out = {
            "metric": "Mrays/s", "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / steps, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name(args.workload, job.sc) + f", {width}x{height}, {BOUNCES} bounces, {FRAMES_PER_STEP} frames "
                                   f"(= {FRAMES_PER_STEP} spp) per step, {args.steps} steps = {args.steps * FRAMES_PER_STEP} spp",
                       "frames_per_step": FRAMES_PER_STEP,
                       "triangles": int(len(job.sc.triangles)), "bvh_nodes": int(len(job.sc.nodes)),
                       "image": [width, height], "max_bounces": BOUNCES,
                       "parallelism": (f"tile-split x{world} (8-row blocks, round robin), one process per GPU, scene replicated, "
                                       "one RCCL gather at the end") if world > 1 else
                                      (f"tile-split x{n_gpus} (8-row blocks, round robin) inside ONE process: device group (mi3pt_create_group), "
                                       "scene replicated, one strided peer copy per member as the gather") if group_devices else "single GPU",
                       "per_rank": m["per_rank"], "host_threads_per_rank": m["host_threads"],
                       **({"tile": args.tile} if args.tile else {}),
                       "rays_per_step": rays // steps,
                       "frames_per_launch": m["frames_per_launch"],
                       "scheduling": "consecutive frames are batched into one persistent launch over (frame, tile) "
                                     "jobs; batches alternate between two streams; one ordered multi-frame "
                                     "accumulate per batch on the main stream"},
        }
        if selftest:
            out["rccl_selftest"] = "world of ONE over the real backend (torch.distributed 'nccl' = RCCL): process group, barrier, gather of the device-resident accumulation image on the context's stream, all_reduce / all_gather of the timings"
        if "gather_verified" in m:
            # the gathered image against shares rendered again on rank 0 (measure(): outside the timed region)
            out["gather_verified"] = m["gather_verified"]
            out["gather_check"] = dict(m["gather_check"], what="a SELF-CONSISTENCY check of the gather and the de-interleave: rank 0 rendered the listed ranks' shares "
                                       "again with the same library, alone (every row, every frame of the job incl. warm-up), and compared them bit for bit with "
                                       "the gathered buffers; its own rows with its own image.  It proves that what arrived is what those ranks rendered and where "
                                       "it belongs -- not that the kernels are right: that is what the parity tests hold against the oracle (tests/test_gpu_configs.py, "
                                       "tests/test_tiles_gloo.py)")
    job.ctx.close()

    # ---- secondary workloads (N = 1): BASELINE.json configs[1], the default demo mesh; and the headline scene seen from
    # close up (the model fills the frame: deep walks on every pixel, next to the sky-and-floor-dominated stated view)
    if world == 1 and args.workload == "dragon" and not args.no_also and not args.tile:
        out["also"] = {}
        for name in ("demo", "closeup"):
            a = measure(name, args.steps, args.warmup, gather=False)
            c = a["counters"]
            out["also"][name] = {"value": round(a["total"]["rays"] / a["elapsed"] / 1e6, 3), "unit": "Mrays/s",
                                 "ms_per_step": round(a["elapsed"] * 1e3 / max(args.steps, 1), 4),
                                 "workload": workload_name(name, a["job"].sc) + f", {width}x{height}, {BOUNCES} bounces",
                                 "kernel_ms": round(a["kernel_ms"], 4), "frames_per_launch": a["frames_per_launch"], "variant": a["variant"],
                                 "rays_per_pixel": round(c["rays"] / max(c["pixels"], 1), 2),
                                 "box_tests_per_ray": round(c["box_tests"] / max(c["rays"], 1), 2),
                                 "tri_tests_per_ray": round(c["tri_tests"] / max(c["rays"], 1), 2)}
            a["job"].ctx.close()
        # the demo scene with the reference's whole render(): raytrace + accumulate + fullscreen (de-noise, ACES, 8-bit canvas)
        # on every frame, canvas semantics as in renderer.ts:379-390 -- half the steps (a frame with its passes takes 2.5x as long)
        psteps = max(args.steps // 2, 1)
        a = measure("demo", psteps, min(args.warmup, 2), gather=False, present=True)
        out["also"]["demo_presenting_every_frame"] = {
            "value": round(a["total"]["rays"] / a["elapsed"] / 1e6, 3), "unit": "Mrays/s",
            "ms_per_frame": round(a["elapsed"] * 1e3 / (psteps * FRAMES_PER_STEP), 4), "steps": psteps,
            "workload": workload_name("demo", a["job"].sc) + f", {width}x{height}, {BOUNCES} bounces; every frame: raytrace, accumulate and "
                        "fullscreen pass (de-noise + ACES + RGBA8 canvas), MI3PT_PRESENT_EXACT",
            "presenting_frames_per_raytrace_launch": a["job"].ctx.get_option(capi.OPT_PRESENT_DEPTH)}
        a["job"].ctx.close()

    # ---- the forest leg (N = 1): config 5's scene -- 10 M triangles, 2.08 GB, the only one larger than the 256 MiB
    # Infinity Cache, i.e. the one on which "HBM GB/s against peak" is the question -- at 1920x1080, with its own counter passes
    forest = None
    if world == 1 and args.workload == "dragon" and not args.no_forest and not args.tile and not args.image:
        forest = measure("forest", FOREST_STEPS, FOREST_WARMUP, gather=False, size=FOREST_IMAGE)
        forest["job"].ctx.close()

    if rank == 0:
        # ---- roofline: counter passes of this very command line (children; the timed job is over)
        log = []
        pmc, source = {}, None
        key = {"workload": args.workload, "image": [width, height], "frames_per_launch": m["frames_per_launch"],
               "variant": args.variant, "n_gpus": n_gpus}
        if world == 1 and not group_devices and not args.no_pmc and not under_profiler():
            try:
                pmc = collect_pmc(args, m["launches"], log)
            except Exception as e:          # noqa: BLE001 -- a profiler problem must not lose the measurement
                log.append(f"pmc: {type(e).__name__}: {e}")
            if pmc:
                source = "live: rocprofv3 --pmc child runs of this command line, mean over the timed launches"
        if "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
            e = traffic_from_file(key)
            if e:
                pmc = dict(e["counters"])
                source = "profiles/traffic.json entry for this launch shape (" + e.get("source", "") + ")"
        out["roofline"] = roofline_block(m, pmc, source, capi_num_cus())
        if n_gpus > 1:
            out["roofline"]["scope"] = (f"ONE GPU of the {n_gpus}: rank 0's raytrace kernel (its live HIP-event times; counters per launch from the "
                                        "same split measured rank by rank on one GPU -- a rank cannot run rocprofv3 passes of its own inside the job)")
        if log:
            out["roofline"]["pmc_log"] = log
        if forest is not None:
            flog, fpmc, fsource = [], {}, None
            if not args.no_pmc and not under_profiler():
                try:
                    fpmc = collect_pmc(args, forest["launches"], flog, workload="forest", steps=FOREST_STEPS, warmup=FOREST_WARMUP, timeout=240)
                except Exception as e:          # noqa: BLE001
                    flog.append(f"pmc: {type(e).__name__}: {e}")
                if fpmc:
                    fsource = "live: rocprofv3 --pmc child runs (bench.py --inner-pmc --workload forest), mean over the timed launches"
            if "FETCH_SIZE" not in fpmc or "WRITE_SIZE" not in fpmc:
                e = traffic_from_file({"workload": "forest", "image": list(FOREST_IMAGE), "frames_per_launch": forest["frames_per_launch"],
                                       "variant": args.variant, "n_gpus": 1})
                if e:
                    fpmc = dict(e["counters"])
                    fsource = "profiles/traffic.json entry for this launch shape (" + e.get("source", "") + ")"
            fj = forest["job"]
            out["forest"] = {"value": round(forest["total"]["rays"] / forest["elapsed"] / 1e6, 3), "unit": "Mrays/s",
                             "ms_per_step": round(forest["elapsed"] * 1e3 / FOREST_STEPS, 4), "steps": FOREST_STEPS, "warmup": FOREST_WARMUP,
                             "workload": workload_name("forest", fj.sc) + f", {FOREST_IMAGE[0]}x{FOREST_IMAGE[1]}, {BOUNCES} bounces "
                                         "(BASELINE.json config 5's scene; its own image, 3840x2160 on 8 GPUs, is a parity-test case)",
                             "scene_bytes": int(len(fj.sc.triangles) * 112 + len(fj.sc.nodes) * 48),
                             "roofline": roofline_block(forest, fpmc, fsource, capi_num_cus())}
            fr = out["forest"]["roofline"]
            if fr.get("hbm", {}).get("achieved") is not None:
                fr["real_bound"] = ("per-wave latency of dependent gathers: this 2 GB scene does not fit the Infinity Cache; its L2 misses are divergent 64-byte sector "
                                    f"requests at {fr['hbm']['achieved']:.0f} GB/s = {fr['hbm']['frac_of_gather_roof']['from_hbm']:.2f} of what a kernel that does nothing but such "
                                    f"gathers reaches out of HBM and {fr['hbm']['frac_of_gather_roof']['from_infinity_cache']:.2f} of what it reaches out of the Infinity Cache "
                                    "(profiles/r05_fetch_calibration.log); a fabric request is outstanding for ~650 L2 cycles on average (TCC_EA0_RDREQ_LEVEL / "
                                    "TCC_EA0_RDREQ, profiles/r05_e_fabric_latency.log) -- the same as on the cache-resident 181 MB scene and half of what the "
                                    "saturated gather probe shows -- so the memory system is not saturated: every node step waits for the slowest of its lanes' "
                                    "round trips, and its waves wait for memory more than they issue; frac is the vector-ALU lane-op fraction, as for the other workloads")
            if flog:
                out["forest"]["roofline"]["pmc_log"] = flog
        if args.tile:
            out["config"]["parallelism"] = f"EXPERIMENT: rank {args.tile} of a tile split rendered alone on one GPU, no gather"
        if world == 1 and not args.no_cpu_baseline and not args.tile:
            out["cpu_baseline"] = cpu_baseline(job)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and out is not None and out.get("gather_verified") is False:
        raise SystemExit("bench.py: the gathered image differs from the shares rendered again on rank 0 (gather_check in the line above)")


def capi_num_cus():
    import torch
    try:
        return int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
    except Exception:       # noqa: BLE001
        return 256


if __name__ == "__main__":
    main()
