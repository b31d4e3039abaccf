import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import bench
from mi3pt_host import capi
capi.load_library()
torch.cuda.set_device(0)
for use_torch_stream, bind in ((True, True), (True, False), (False, False)):
    stream = torch.cuda.Stream() if use_torch_stream else None
    job = bench.Job("dragon", 1920, 1080, 3, 8, 0, 0, stream.cuda_stream if stream else None)
    ctx = job.ctx
    if bind:
        accum = torch.zeros((ctx.local_rows, 1920, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.bind_accumulation(accum.data_ptr(), accum.numel() * 4)
    per_launch = bench.frames_per_launch(ctx.batch_capacity())
    job.frames(80, per_launch); ctx.sync()
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.frames(320, per_launch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"torch_stream={use_torch_stream} bind={bind} rep {rep}: submit {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms, stats {ctx.raytrace_launch_stats(reset=True)}", flush=True)
    if bind: ctx.bind_accumulation(None, 0)
    ctx.close()
