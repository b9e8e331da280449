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


class ToyEngine:
    """CPU stand-in with RayRenderEngine's shard interface: fake_render per ray, and — like the real path — an
    ``expected_depth`` that is clipped to the [min, max] of a per-ray quantity over each ``chunk``-ray chunk of the frame."""

    def __init__(self, chunk):
        self.chunk = chunk

    @staticmethod
    def _mid(o, d):
        return (o * 3.0 + d).sum(-1, keepdim=True) * 0.5

    def render(self, o, d):  # the unsharded frame
        out = fake_render(o, d)
        mid = self._mid(o, d)
        ed = out["expected_depth"].clone()
        for i in range(0, o.shape[0], self.chunk):
            m = mid[i:i + self.chunk]
            ed[i:i + self.chunk] = ed[i:i + self.chunk].clamp(m.min() * 0.9, m.max() * 0.9)
        out["expected_depth"] = ed
        return out

    def render_shard(self, o, d, start, frame_rays):
        n = o.shape[0]
        out = fake_render(o, d)  # expected_depth unclipped
        mid = self._mid(o, d)
        n_chunks = -(-frame_rays // self.chunk)
        bounds = torch.tensor([[float("inf"), float("-inf")]] * n_chunks)
        if n:
            for c in range(start // self.chunk, (start + n - 1) // self.chunk + 1):
                p0, p1 = max(c * self.chunk, start), min((c + 1) * self.chunk, frame_rays, start + n)
                m = mid[p0 - start:p1 - start]
                bounds[c, 0], bounds[c, 1] = m.min() * 0.9, m.max() * 0.9
        return out, bounds

    def apply_depth_bounds(self, out, start, bounds):
        n = out["expected_depth"].shape[0]
        for c in range(start // self.chunk, (start + n - 1) // self.chunk + 1) if n else ():
            i, j = max(c * self.chunk, start) - start, min((c + 1) * self.chunk, start + n) - start
            out["expected_depth"][i:j].clamp_(bounds[c, 0], bounds[c, 1])


def _fine_worker(rank, world, port, h, w, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o, d, _ = synthetic.orbit_camera_rays(h, w, view=2)
        eng = ToyEngine(chunk)
        full = D.render_frame_sharded_fine(eng, o, d, align=4)
        want = eng.render(o.reshape(-1, 3), d.reshape(-1, 3))
        ok = all(torch.equal(full[k].reshape(-1, full[k].shape[-1]), want[k]) for k in D.OUTPUT_KEYS)
        q.put((rank, ok, tuple(full["rgb"].shape)))
    finally:
        dist.destroy_process_group()


# world sizes 2 and 3; chunks split between ranks, a short last chunk, one chunk for the whole frame, fewer tiles than ranks
@pytest.mark.parametrize("h,w,chunk,world", [(16, 12, 50, 2), (16, 12, 50, 3), (7, 5, 64, 3), (16, 12, 48, 2), (1, 3, 64, 3)])
def test_sub_chunk_sharded_frame_restores_the_chunk_wide_depth_clip(h, w, chunk, world):
    """distributed.render_frame_sharded_fine: even ray shards that cut through the reference's chunks, the per-chunk clip bounds
    joined by ONE all-reduce, the frame equal to the unsharded one (the toy engine clips per chunk like the real one)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fine_worker, args=(r, world, port, h, w, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, shape in res:
        assert ok, f"rank {rank} mismatch"
        assert shape == (h, w, 3)


def test_ray_blocks_are_even_whatever_the_chunk_size():
    for n in (640000, 2073600, 100, 64, 5):
        for world in (1, 2, 3, 8):
            blocks = [D.ray_block(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            assert all(b[0] % 64 == 0 for b in blocks if b[0] < n)
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) < 128  # one 64-ray tile, plus the frame's ragged last tile
    # the headline frame over 8 ranks: 80 000 rays each (whole 65 536-ray chunks give 2, 2, 1, 1, 1, 1, 1, 1 chunks)
    assert [b[1] - b[0] for b in (D.ray_block(640000, r, 8) for r in range(8))] == [80000] * 8


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


class SplitRecordingEngine(ToyEngine):
    """ToyEngine + the engine's sample-split surface: proposes k from the size of a rank's run and records what render_shard got"""

    def __init__(self, chunk):
        super().__init__(chunk)
        self.asked, self.got, self.planes = [], [], []

    def shard_sample_split(self, shard_rays):
        self.asked.append(shard_rays)
        return 1 + shard_rays % 5

    def render_shard(self, o, d, start, frame_rays, sample_split=None, nears=None, fars=None):
        self.got.append(sample_split)
        self.planes.append(None if nears is None else (tuple(nears.shape), tuple(fars.shape)))
        return super().render_shard(o, d, start, frame_rays)


def _split_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o, d, _ = synthetic.orbit_camera_rays(16, 12, view=1)
        n = 16 * 12
        eng = SplitRecordingEngine(50)
        D.render_frame_sharded_fine(eng, o, d, align=4, sample_split="shard")  # opt-in: the k that suits ONE rank's run
        D.render_frame_sharded_fine(eng, o, d, align=4)                        # default: the frame's own choice, nothing passed on
        D.render_frame_sharded_fine(eng, o, d, align=4, sample_split=3, nears=torch.zeros(16, 12, 1), fars=torch.ones(16, 12, 1))
        r0, r1 = D.ray_block(n, rank, world, 4)
        q.put((rank, eng.asked, eng.got, eng.planes, D.ray_block(n, 0, world, 4)[1], r1 - r0))
    finally:
        dist.destroy_process_group()


def test_fine_sharding_hands_every_rank_the_same_sample_split():
    """render_frame_sharded_fine: sample_split="shard" asks the engine ONCE per frame for the k of RANK 0's run size (so every rank
    passes the same k, whatever its own run), the default (None) passes nothing (the engine's frame-level choice: the sharded frame
    is the default single-device frame bit for bit, whatever the world size — ADVICE r5), an integer is forced; per-ray
    planes are sliced like the rays."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    ks = set()
    for rank, asked, got, planes, run0, mine in res:
        assert asked == [run0]
        assert got == [1 + run0 % 5, None, 3]
        assert planes == [None, None, ((mine,), (mine,))]
        ks.add(got[0])
    assert len(ks) == 1


def _broadcast_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig

        cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=8, log2_hashmap_size=10)
        model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=3)
        # rank 0 holds "the checkpoint" (a counter-hash fill); every other rank starts from its own random initialisation
        torch.manual_seed(100 + rank)
        if rank == 0:
            synthetic.fill_model_(model, "stress")
        else:
            with torch.no_grad():
                for p in model.parameters():
                    p.normal_()
        calls = []
        real = dist.broadcast

        def counting(t, *a, **k):
            calls.append(t.numel() * t.element_size())
            return real(t, *a, **k)

        dist.broadcast = counting
        try:
            moved = D.broadcast_model_(model, src=0)
        finally:
            dist.broadcast = real
        want = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=3)
        synthetic.fill_model_(want, "stress")
        got, ref = dict(model.named_parameters()), dict(want.named_parameters())
        ok = set(got) == set(ref) and all(torch.equal(got[k], ref[k]) for k in ref)
        bufs = dict(model.named_buffers())
        ok = ok and all(torch.equal(bufs[k], b) for k, b in want.named_buffers())
        n_param_bytes = sum(p.numel() * p.element_size() for p in model.parameters()) + sum(
            b.numel() * b.element_size() for b in model.buffers())
        q.put((rank, ok, moved, n_param_bytes, len(calls)))
    finally:
        dist.destroy_process_group()


def test_broadcast_model_makes_every_rank_hold_rank_zeros_weights():
    """weights replicated through ONE broadcast at load (distributed.broadcast_model_): only rank 0 has the weights;
    after the call every parameter and buffer of both ranks equals rank 0's bit for bit, and it took one collective per dtype,
    not one per tensor."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_broadcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, moved, total, n_calls in res:
        assert ok, f"rank {rank}: weights differ from rank 0's"
        assert moved == total and n_calls <= 3, (moved, total, n_calls)


def test_bench_gpus_flag_against_the_launchers_world_size(monkeypatch):
    """bench.py: --gpus N without a launcher asks for N ranks to be started (-1); under a launcher it must equal WORLD_SIZE — a
    contradiction is an error, never a silent one-GPU run."""
    import argparse

    import bench

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    for given, want in ((None, 1), (1, 1), (2, -1), (8, -1)):
        a = argparse.Namespace(gpus=given)
        assert bench.resolve_world(a) == want
        assert a.gpus == (given or 1)
    monkeypatch.setenv("WORLD_SIZE", "4")
    a = argparse.Namespace(gpus=None)
    assert bench.resolve_world(a) == 4 and a.gpus == 4
    assert bench.resolve_world(argparse.Namespace(gpus=4)) == 4
    for bad in (1, 2, 8):
        with pytest.raises(SystemExit):
            bench.resolve_world(argparse.Namespace(gpus=bad))
