"""CPU, world_size 2, gloo: the ray-shard + pixel all-gather plumbing used for N > 1 GPUs (the render itself is a
stand-in pure function here — HIP kernels need a GPU; what is covered is partitioning, packing and the collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from thermo_nerf_amd import distributed as D
from thermo_nerf_amd import synthetic


def fake_render(o, d):
    """deterministic per-ray function with the engine's output signature"""
    s = (o * 3.0 + d).sum(-1, keepdim=True)
    out = {"rgb": torch.cat([torch.sin(s), torch.cos(s), torch.sin(2 * s)], dim=1)}
    for i, k in enumerate(D.OUTPUT_KEYS[1:]):
        out[k] = s * (i + 1)
    return out


def _worker(rank, world, port, h, w, q, chunk=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o, d, _ = synthetic.orbit_camera_rays(h, w, view=1)
        full = D.render_frame_sharded(fake_render, o, d, chunk=chunk)
        want = fake_render(o.reshape(-1, 3), d.reshape(-1, 3))
        ok = all(torch.equal(full[k].reshape(-1, full[k].shape[-1]), want[k]) for k in D.OUTPUT_KEYS)
        lo, hi = D.reduce_depth_bounds(torch.tensor(float(rank + 1)), torch.tensor(float(rank + 1)))
        ok = ok and float(lo) == 1.0 and float(hi) == float(world)
        q.put((rank, ok, tuple(full["rgb"].shape)))
    finally:
        dist.destroy_process_group()


# even and uneven row blocks; chunk-aligned sharding: even chunk counts, a short last chunk, fewer chunks than ranks
@pytest.mark.parametrize("h,w,chunk", [(16, 12, None), (7, 5, None), (16, 12, 48), (16, 12, 50), (7, 5, 64)])
def test_row_sharded_frame_gathers_back(h, w, chunk):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, h, w, q, chunk)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, shape in res:
        assert ok, f"rank {rank} mismatch"
        assert shape == (h, w, 3)


def test_row_blocks_partition_the_image():
    for h in (1, 7, 800, 1080):
        for world in (1, 2, 3, 8):
            blocks = [D.row_block(h, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == h
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_chunk_blocks_partition_the_frame_on_chunk_boundaries():
    for n, chunk in ((640000, 65536), (2073600, 65536), (100, 64), (64, 64), (5, 64)):
        for world in (1, 2, 3, 8):
            blocks = [D.chunk_block(n, chunk, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            assert all(b[0] % chunk == 0 for b in blocks if b[0] < n)  # every shard starts on a reference chunk boundary
            per = [-(-(b[1] - b[0]) // chunk) for b in blocks]
            assert max(per) - min(per) <= 1


def test_pack_unpack_roundtrip():
    o, d, _ = synthetic.orbit_camera_rays(4, 4)
    out = fake_render(o.reshape(-1, 3), d.reshape(-1, 3))
    back = D.unpack_outputs(D.pack_outputs(out))
    assert D.pack_outputs(out).shape == (16, 9)
    for k in D.OUTPUT_KEYS:
        assert torch.equal(back[k], out[k])


def _pipeline_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 50
        pipe = D.PipelinedFrameGather(n, world, "cpu")
        ok, slots = True, []
        for frame in range(5):  # more frames than buffers: slots are recycled only after their collective finished
            o = torch.full((n, 3), float(frame)) + rank
            d = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3) / 100.0
            slots.append((pipe.submit(fake_render(o, d)), frame))
        pipe.finish()
        for k, frame in slots[-2:]:  # the two frames still resident
            got = pipe.frames(k, world)
            for r in range(world):
                o = torch.full((n, 3), float(frame)) + r
                d = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3) / 100.0
                ok = ok and torch.equal(got[r], D.pack_outputs(fake_render(o, d)))
        q.put((rank, ok, [k for k, _ in slots]))
    finally:
        dist.destroy_process_group()


def test_pipelined_frame_gather_double_buffering():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, slots in res:
        assert ok, f"rank {rank}: gathered frames differ"
        assert slots == [0, 1, 0, 1, 0]
