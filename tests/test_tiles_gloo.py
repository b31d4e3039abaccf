"""Multi-GPU path: world_size 2 (and 3) over gloo.  Each rank renders its round-robin 8-row
blocks, the HDR accumulation buffers are gathered to rank 0 exactly as bench.py does over RCCL,
and the de-interleaved image must equal the single-rank image bit for bit.  On the CPU (no GPU in
the build container) the oracle stands in for the device -- that checks the deal / gather /
de-interleave arithmetic; on the GPU box the same flow runs with the HIP path rendering."""
import os
import socket
import sys

import numpy as np
import pytest

import ptcommon as pc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, block, outq):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "webgpu-pathtracer_amd", "py"), os.path.join(root, "oracle"), here):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import pt_oracle as orc
    from mi3pt_host import capi, scenes
    from mi3pt_host.tiles import deinterleave_rows
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.demo_scene()
    sc.build_bvh(nthreads=2)
    env = scenes.synthetic_env()
    osc = orc.OracleScene(sc.triangles, sc.material_bytes, sc.nodes, env)
    rows = capi.tile_local_rows(h, rank, world, block)
    acc = np.zeros((rows, w, 4), np.float32)
    for frame in (2, 3):
        img, _ = orc.raytrace(osc, pc.rt_uniforms(sc, w, h, frame=frame, bounces=3).tobytes(), w, h, rank, world, block)
        acc = orc.accumulate(pc.acc_uniforms(w, h, frame).tobytes(), w, h, img, acc, rank, world, block)
    max_rows = max(capi.tile_local_rows(h, r, world, block) for r in range(world))      # (the deal goes back and forth: rank 0 need not hold the most rows)
    send = torch.zeros((max_rows, w, 4))
    send[:rows] = torch.from_numpy(acc)
    gathered = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, gathered, dst=0)
    if rank == 0:
        whole = deinterleave_rows([g.numpy() for g in gathered], h, world, block)
        outq.put(whole)
    dist.barrier()
    dist.destroy_process_group()


def _gpu_worker(rank, world, port, w, h, block, frames, outq):
    """The same flow with the DEVICE path doing the rendering: every rank owns a context on the
    one GPU of the box with its tile set, accumulates into a torch tensor bound as the
    accumulation image (bench.py's arrangement), and the tensors are gathered over gloo."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "webgpu-pathtracer_amd", "py"), here):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from mi3pt_host import capi, scenes
    from mi3pt_host.tiles import deinterleave_rows
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.demo_scene()
    sc.build_bvh(nthreads=2)
    ctx = capi.Context(0)
    ctx.set_option(capi.OPT_BATCH, 4)         # 4 x world frames per launch: the 20 frames take several launches
    pc.upload_scene(ctx, sc, scenes.synthetic_env())
    ctx.set_tile(rank, world, block)
    ctx.resize(w, h)
    accum = torch.zeros((ctx.local_rows, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.bind_accumulation(accum.data_ptr(), accum.numel() * 4)
    for frame in frames:
        pc.gpu_frame(ctx, pc.rt_uniforms(sc, w, h, frame=frame, bounces=5), pc.acc_uniforms(w, h, frame),
                     capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
    ctx.sync()
    max_rows = max(capi.tile_local_rows(h, r, world, block) for r in range(world))      # (the deal goes back and forth: rank 0 need not hold the most rows)
    send = torch.zeros((max_rows, w, 4))
    send[: ctx.local_rows] = accum.cpu()
    gathered = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, gathered, dst=0)
    if rank == 0:
        outq.put(deinterleave_rows([g.numpy() for g in gathered], h, world, block))
    dist.barrier()
    ctx.bind_accumulation(None, 0)
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_tile_split_on_the_device_gathers_to_the_whole_image(gpu_ctx, demo, env, world):
    """world_size 2 and 3 over gloo, the HIP path rendering (one context per rank on the one GPU;
    20 frames in launches of 4 x world): the gathered, de-interleaved HDR image equals the image a
    single 1-rank context renders, bit for bit."""
    import torch.multiprocessing as mp
    from mi3pt_host import capi
    w, h, block, frames = 200, 119, 8, list(range(2, 22))
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_gpu_worker, args=(r, world, port, w, h, block, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    whole = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ctx = gpu_ctx
    pc.upload_scene(ctx, demo, env)
    ctx.set_tile(0, 1, 8)
    ctx.resize(w, h)
    for frame in frames:
        pc.gpu_frame(ctx, pc.rt_uniforms(demo, w, h, frame=frame, bounces=5), pc.acc_uniforms(w, h, frame),
                     capi.SUBMIT_RAYTRACE | capi.SUBMIT_ACCUMULATE)
    want = ctx.read_texture(capi.TEX_ACCUMULATION)
    assert pc.same_bits(whole, want), pc.describe_diff(whole, want)
    ctx.resize(64, 64)


@pytest.mark.parametrize("w,h,block", [(48, 40, 8), (40, 37, 5)])
def test_two_rank_tile_split_gathers_to_the_whole_image(orc, demo, env, w, h, block):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, block, q)) for r in range(2)]
    for p in procs:
        p.start()
    whole = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    osc = pc.oracle_scene(orc, demo, env)
    acc = np.zeros((h, w, 4), np.float32)
    for frame in (2, 3):
        img, _ = orc.raytrace(osc, pc.rt_uniforms(demo, w, h, frame=frame, bounces=3).tobytes(), w, h)
        acc = orc.accumulate(pc.acc_uniforms(w, h, frame).tobytes(), w, h, img, acc)
    assert pc.same_bits(whole, acc), pc.describe_diff(whole, acc)


def test_deinterleave_is_the_inverse_of_the_row_deal():
    from mi3pt_host import capi
    from mi3pt_host.tiles import deinterleave_rows, local_rows_of
    h, w = 70, 3
    img = np.arange(h * w * 4, dtype=np.float32).reshape(h, w, 4)
    for world, block in ((1, 8), (2, 8), (3, 5), (8, 8), (4, 16)):
        parts = []
        for r in range(world):
            rows = local_rows_of(h, r, world, block)
            assert len(rows) == capi.tile_local_rows(h, r, world, block)
            pad = np.zeros((max(capi.tile_local_rows(h, q, world, block) for q in range(world)), w, 4), np.float32)
            pad[:len(rows)] = img[rows]
            parts.append(pad)
        assert np.array_equal(deinterleave_rows(parts, h, world, block), img)
        # the deal is the LIBRARY's (tiles.local_rows_of asks it row by row): every image row has exactly one owner, mi3pt_tile_owner names
        # it, and a rank's rows come in ascending order, in whole blocks of `block` rows except at the image's end
        assert sorted(y for r in range(world) for y in local_rows_of(h, r, world, block)) == list(range(h))
        for r in range(world):
            rows = local_rows_of(h, r, world, block)
            assert rows == sorted(rows) and all(capi.tile_owner(y, world, block) == r for y in rows)
            assert all(rows[i + 1] == rows[i] + 1 for i in range(len(rows) - 1) if (i + 1) % block)
