"""GPU: BASELINE config 4 — one camera-path frame at 1920x1080 through the real engine, unsharded and ray-sharded.

* a full-size frame of the reference's own camera path (tests/golden/camera_path_facade_2.json, 96 poses): size-independent
  properties + the oracle on a strided sample;
* ``distributed.render_frame_sharded`` driving the REAL ``RayRenderEngine``: world size 1 on RCCL ("nccl"), and two ranks
  sharing the one GPU of the box on gloo (RCCL refuses two ranks on one device), row-block and chunk-aligned, against the
  unsharded frame — the chunk-aligned form must equal it bit for bit, expected depth included.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
S = 48  # the reference's default num_nerf_samples_per_ray (config 4 renders with the trained model's config)
CHUNK = 1 << 16  # REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:30 eval_num_rays_per_chunk


def _model(dev=DEV):
    from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic

    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=S, eval_num_rays_per_chunk=CHUNK)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    sd = synthetic.model_state_dict_cpu(model)
    return model.to(dev).eval(), sd, cfg


def _frame_rays(index: int, dev=DEV, scale: float = 0.45):
    """Rays of pose ``index`` of the reference's camera path at its native 1920x1080, camera centres scaled into the unit
    scene box (the path was recorded around a real scene; the synthetic weights live in [-1,1]^3)."""
    from thermo_nerf_amd.cameras import get_path_from_json

    import json

    cams = get_path_from_json(json.load(open(os.path.join(GOLDEN, "camera_path_facade_2.json"))))
    assert len(cams) == 96 and cams.height == 1080 and cams.width == 1920  # REF tests/test_renderer.py:65-69
    c2w = cams.camera_to_worlds.clone()
    c2w[:, :3, 3] *= scale / c2w[:, :3, 3].norm(dim=-1).max()
    cams.camera_to_worlds = c2w
    rb = cams.generate_rays(index, device=dev)
    return rb.origins, rb.directions  # [1080,1920,3]


def test_full_1080p_frame_properties():
    from oracle import hotpath as H
    from tests import helpers
    from thermo_nerf_amd.engine import RayRenderEngine

    model, sd, cfg = _model()
    o3, d3 = _frame_rays(17)
    assert o3.shape == (1080, 1920, 3)
    o, d = o3.reshape(-1, 3).contiguous(), d3.reshape(-1, 3).contiguous()
    n = o.shape[0]
    assert n == 2073600
    whole = RayRenderEngine(model, chunk=n)
    a = {k: v.clone() for k, v in whole.render(o, d).items()}
    b = whole.render(o, d)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k}: not idempotent"
        assert torch.isfinite(a[k]).all(), k
    c = RayRenderEngine(model, chunk=CHUNK).render(o, d)  # the reference's chunking, two streams
    torch.cuda.synchronize()
    for k in ("rgb", "thermal", "accumulation", "depth", "prop_depth_0", "prop_depth_1"):
        assert torch.equal(a[k], c[k]), f"{k}: depends on chunking"
    # through the plugin surface: get_outputs_for_camera_ray_bundle == the engine at eval_num_rays_per_chunk
    from thermo_nerf_amd import RayBundle

    e = model.get_outputs_for_camera_ray_bundle(RayBundle(origins=o3, directions=d3))
    assert e["rgb"].shape == (1080, 1920, 3) and e["thermal"].shape == (1080, 1920, 1)
    for k in c:
        assert torch.equal(e[k].reshape(n, -1), c[k]), k
    # the oracle on a strided sample of the frame
    idx = torch.linspace(0, n - 1, 1536).long()
    want = H.get_outputs(sd, o[idx.to(DEV)].cpu(), d[idx.to(DEV)].cpu(), None, helpers.oracle_config(cfg))
    for k, tol in (("rgb", 1e-4), ("thermal", 1e-4), ("accumulation", 2e-5)):
        err = (a[k][idx.to(DEV)].cpu() - want[k]).abs()
        assert err.mean().item() <= tol and err.max().item() <= 20 * tol, (k, err.mean().item(), err.max().item())


def test_camera_path_poses_through_the_plugin_surface_match_the_oracle():
    """BASELINE config 4's loop [REF scripts/render_video_script.py:59-91, render/renderer.py:160-201] on three poses of the
    reference's camera path (first, middle, last): Cameras.generate_rays -> get_outputs_for_camera_ray_bundle at the reference's
    chunk size -> [1080,1920,C] frames, each against the CPU oracle on a strided sample of its rays (bench.py times all 96)."""
    from oracle import hotpath as H
    from tests import helpers
    from thermo_nerf_amd import RayBundle

    model, sd, cfg = _model()
    for pose in (0, 48, 95):
        o3, d3 = _frame_rays(pose)
        out = model.get_outputs_for_camera_ray_bundle(RayBundle(origins=o3, directions=d3))
        torch.cuda.synchronize()
        assert out["rgb"].shape == (1080, 1920, 3) and out["thermal"].shape == (1080, 1920, 1) and out["depth"].shape == (1080, 1920, 1)
        n = 1080 * 1920
        idx = torch.linspace(0, n - 1, 768).long().to(DEV)
        want = H.get_outputs(sd, o3.reshape(-1, 3)[idx].cpu(), d3.reshape(-1, 3)[idx].cpu(), None, helpers.oracle_config(cfg))
        for k, tol in (("rgb", 1e-4), ("thermal", 1e-4), ("accumulation", 2e-5)):
            err = (out[k].reshape(n, -1)[idx].cpu() - want[k]).abs()
            assert err.mean().item() <= tol and err.max().item() <= 20 * tol, (pose, k, err.mean().item(), err.max().item())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharded_frame_world_size_1_on_rccl():
    """The N = 1 degenerate case on the real backend: init RCCL, shard (one block), all-gather, compare."""
    from thermo_nerf_amd import distributed as D
    from thermo_nerf_amd.engine import RayRenderEngine

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        model, _, _ = _model()
        # the load-time weight broadcast on the real backend (RCCL takes device tensors of every dtype the model holds: fp32
        # parameters, the int64 / fp32 buffers): 78 MB, parameters unchanged at world size 1
        before = {k: v.clone() for k, v in model.state_dict().items()}
        moved = D.broadcast_model_(model, src=0)
        torch.cuda.synchronize()
        assert 70e6 < moved < 90e6 and all(torch.equal(v, before[k]) for k, v in model.state_dict().items())
        o3, d3 = _frame_rays(3)
        eng = RayRenderEngine(model, chunk=CHUNK)
        want = eng.render(o3.reshape(-1, 3).contiguous(), d3.reshape(-1, 3).contiguous())
        want = {k: v.clone() for k, v in want.items()}
        for chunk in (None, CHUNK):
            got = D.render_frame_sharded(eng.render, o3, d3, device=torch.device(DEV), chunk=chunk)
            torch.cuda.synchronize()
            for k in D.OUTPUT_KEYS:
                assert got[k].shape[:2] == (1080, 1920)
                assert torch.equal(got[k].reshape(want[k].shape), want[k]), (k, chunk)
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from thermo_nerf_amd import distributed as D
        from thermo_nerf_amd.engine import RayRenderEngine

        model, _, _ = _model()
        o3, d3 = _frame_rays(40)
        eng = RayRenderEngine(model, chunk=CHUNK)
        res = {}
        for name, chunk in (("rows", None), ("chunks", CHUNK)):
            got = D.render_frame_sharded(eng.render, o3, d3, device=torch.device(DEV), chunk=chunk)
            torch.cuda.synchronize()
            if rank == 0:
                want = eng.render(o3.reshape(-1, 3).contiguous(), d3.reshape(-1, 3).contiguous())
                torch.cuda.synchronize()
                for k in D.OUTPUT_KEYS:
                    res[f"{name}.{k}"] = bool(torch.equal(got[k].reshape(want[k].shape), want[k]))
                ed = (got["expected_depth"].reshape(-1) - want["expected_depth"].reshape(-1)).abs().max().item()
                res[f"{name}.expected_depth_maxdiff"] = ed
        dist.barrier()
        q.put((rank, res, None))
    except Exception as e:  # pragma: no cover - surfaced by the parent
        import traceback

        q.put((rank, {}, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_sharded_frame_two_ranks_sharing_the_gpu():
    """Two processes (gloo) render their shares of one 1080p frame with the real engine on the same GPU and all-gather it.
    Chunk-aligned shards reproduce the single-device frame exactly (the reference clips expected depth per 65 536-ray
    chunk); row-block shards agree on everything but that per-chunk clip."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, r, err in res:
        assert err is None, f"rank {rank}:\n{err}"
    r0 = [r for rank, r, _ in res if rank == 0][0]
    from thermo_nerf_amd import distributed as D

    for k in D.OUTPUT_KEYS:
        assert r0[f"chunks.{k}"], f"chunk-aligned shard differs in {k}"
        if k != "expected_depth":
            assert r0[f"rows.{k}"], f"row-block shard differs in {k}"


@pytest.mark.parametrize("h,w,chunk,world", [(800, 800, CHUNK, 8), (1080, 1920, CHUNK, 8), (300, 400, CHUNK, 3), (250, 300, CHUNK, 2),
                                             (200, 300, 1 << 20, 4)])
def test_sub_chunk_shards_reproduce_the_unsharded_frame(h, w, chunk, world):
    """distributed.ray_block + RayRenderEngine.render_shard / apply_depth_bounds in ONE process: the shards all ``world`` ranks
    would render (even runs of rays cut on multiples of 64, NOT on chunk boundaries: 800x800 over 8 ranks = 80 000 rays each
    where whole chunks give 2, 2, 1, 1, 1, 1, 1, 1), their per-chunk depth bounds min/max-reduced as the all-reduce does, against
    the unsharded frame: bit-equal in all seven outputs, expected depth included.  Frame sizes on both sides of the two
    call-size thresholds of kernel_family="auto" (tn_render_kernel_form) and a frame that is one chunk."""
    from thermo_nerf_amd import distributed as D
    from thermo_nerf_amd import synthetic
    from thermo_nerf_amd.engine import RayRenderEngine

    model, _, _ = _model()
    o3, d3, _ = synthetic.orbit_camera_rays(h, w, view=5)
    o, d = o3.reshape(-1, 3).contiguous().to(DEV), d3.reshape(-1, 3).contiguous().to(DEV)
    n = h * w
    eng = RayRenderEngine(model, chunk=chunk)
    want = {k: v.clone() for k, v in eng.render(o, d).items()}
    blocks = [D.ray_block(n, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n and all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
    assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) < 128  # one tile + the ragged last tile
    shards = []
    for a, b in blocks:
        out, bounds = eng.render_shard(o[a:b].contiguous(), d[a:b].contiguous(), a, n)
        shards.append(({k: v.clone() for k, v in out.items()}, a, bounds.clone()))
    lo = torch.stack([b[:, 0] for _, _, b in shards]).min(dim=0).values
    hi = torch.stack([b[:, 1] for _, _, b in shards]).max(dim=0).values
    bounds = torch.stack([lo, hi], dim=1).contiguous()
    assert torch.isfinite(bounds).all()  # every chunk of the frame was touched by some shard
    for out, a, _ in shards:
        eng.apply_depth_bounds(out, a, bounds)
    torch.cuda.synchronize()
    # the engine's own two forms of the unsharded frame agree as well: chunks fused into one launch pair (default) and one
    # launch pair per chunk
    # (from 81 920 rays per call both passes run lane = ray in either form; below, kernel_family="auto" may pick the
    # ray-per-wave proposal pass for the one fused call and lane = ray for overlapping chunk calls: same tolerance, other bits)
    if n >= 81920:
        per_chunk = RayRenderEngine(model, chunk=chunk, fuse_chunks=False).render(o, d)
        torch.cuda.synchronize()
        for k in D.OUTPUT_KEYS:
            assert torch.equal(per_chunk[k], want[k]), (k, "fused chunks differ from launch-per-chunk")
    for k in D.OUTPUT_KEYS:
        got = torch.cat([out[k] for out, _, _ in shards])
        assert torch.equal(got, want[k]), (k, (got - want[k]).abs().max().item())


def _shards_against_the_frame(eng, o, d, n, world, nears=None, fars=None, sample_split=None):
    """every rank's render_shard + the reduced bounds, concatenated, against eng.render of the whole frame: bit for bit"""
    from thermo_nerf_amd import distributed as D

    want = {k: v.clone() for k, v in eng.render(o, d, nears=nears, fars=fars, sample_split=sample_split).items()}
    shards = []
    for r in range(world):
        a, b = D.ray_block(n, r, world)
        kw = {} if nears is None else {"nears": nears[a:b].contiguous(), "fars": fars[a:b].contiguous()}
        if sample_split is not None:
            kw["sample_split"] = sample_split
        out, bounds = eng.render_shard(o[a:b].contiguous(), d[a:b].contiguous(), a, n, **kw)
        shards.append(({k: v.clone() for k, v in out.items()}, a, bounds.clone()))
    lo = torch.stack([b[:, 0] for _, _, b in shards]).min(dim=0).values
    hi = torch.stack([b[:, 1] for _, _, b in shards]).max(dim=0).values
    bounds = torch.stack([lo, hi], dim=1).contiguous()
    for out, a, _ in shards:
        eng.apply_depth_bounds(out, a, bounds)
    torch.cuda.synchronize()
    for k in D.OUTPUT_KEYS:
        got = torch.cat([out[k] for out, _, _ in shards])
        assert torch.equal(got, want[k]), (k, (got - want[k]).abs().max().item())
    return want


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("h,w", [(200, 250), (180, 250), (250, 400)])
def test_shards_take_the_whole_launch_kernel_form_in_every_precision(h, w, precision):
    """The kernel form of a launch is the LIBRARY's decision (tn_render_kernel_form): with a split-precision blob the field pass
    runs lane = ray from 40 960 rays per call, with exact fp32 (which marches small calls in segments) from 8 192, the proposal pass
    from 24 576 — frames of 45 000, 50 000 and 100 000 rays sit between / above them.  A shard (1/4 of the frame: below every threshold) must run the form of
    the unsharded launch, in every precision (the engine used to re-derive the fp32 thresholds in Python)."""
    from thermo_nerf_amd import _hip, synthetic
    from thermo_nerf_amd.engine import RayRenderEngine

    model, _, _ = _model()
    model.config.mlp_precision = precision
    model.invalidate_prepared()
    n = h * w
    o3, d3, _ = synthetic.orbit_camera_rays(h, w, view=2)
    o, d = o3.reshape(-1, 3).contiguous().to(DEV), d3.reshape(-1, 3).contiguous().to(DEV)
    eng = RayRenderEngine(model, chunk=CHUNK)
    _, _, fld = model._c_structs()
    eng.rc.kernel_family = 0
    form = eng.lib.tn_render_kernel_form(fld, eng.rc, n, 1)
    assert form == (1 if n >= (8192 if precision == "f32" else 40960) else 2)
    assert eng.lib.tn_render_kernel_form(fld, eng.rc, n, 0) == (1 if n >= 24576 else 2)
    assert eng._forms(fld, n, 0)[:2] == (eng.lib.tn_render_kernel_form(None, eng.rc, n, 0), form)
    _shards_against_the_frame(eng, o, d, n, 4)
    assert eng.rc.kernel_family == 0  # render_shard leaves the engine's setting alone


@pytest.mark.parametrize("h,w,world", [(800, 800, 8), (300, 400, 3)])
def test_sample_split_shards_reproduce_the_frame_rendered_with_that_split(h, w, world):
    """Strong scaling of a frame whose per-rank run under-fills the chip (800 x 800 over 8 ranks: 1 250 tiles on 2 048 wave slots):
    the field pass marches every tile in k segments (engine.shard_sample_split(run size): the same k on every rank).  The split
    is a property of the FRAME — the shards equal engine.render(frame, sample_split=k) bit for bit, expected depth included — and
    that frame sits within rounding of the default one."""
    from thermo_nerf_amd import distributed as D
    from thermo_nerf_amd import synthetic
    from thermo_nerf_amd.engine import RayRenderEngine

    model, _, _ = _model()
    n = h * w
    o3, d3, _ = synthetic.orbit_camera_rays(h, w, view=3)
    o, d = o3.reshape(-1, 3).contiguous().to(DEV), d3.reshape(-1, 3).contiguous().to(DEV)
    eng = RayRenderEngine(model, chunk=CHUNK)
    k = eng.shard_sample_split(D.ray_block(n, 0, world)[1])
    assert k > 1
    got = _shards_against_the_frame(eng, o, d, n, world, sample_split=k)
    plain = eng.render(o, d)
    for name in ("rgb", "thermal", "accumulation"):
        assert (got[name] - plain[name]).abs().max().item() <= 3e-6, name
    assert eng.rc.sample_split == 0 and eng.rc.kernel_family == 0


def test_shards_of_a_frame_cut_into_several_launches_on_one_stream():
    """A frame longer than the workspace budget: equal runs of whole chunks (frame_launch_rays), the LAST launch shorter and —
    here — below the lane = ray threshold while the others are above it; a shard's pieces take the form of the launch that
    holds them, on one stream as on two."""
    from thermo_nerf_amd import synthetic
    from thermo_nerf_amd.engine import RayRenderEngine

    model, _, _ = _model()
    h, w = 300, 620  # 186 000 rays, chunk 8192: 23 chunks
    n = h * w
    o3, d3, _ = synthetic.orbit_camera_rays(h, w, view=6)
    o, d = o3.reshape(-1, 3).contiguous().to(DEV), d3.reshape(-1, 3).contiguous().to(DEV)
    for streams in (1, 2):
        eng = RayRenderEngine(model, chunk=8192, streams=streams, max_workspace_bytes=150 << 20)
        L = eng.frame_launch_rays(n)
        assert L % 8192 == 0 and L < n
        pieces = eng._launch_pieces(0, n, n)
        assert len(pieces) >= 2 and pieces[-1][1] == n and all(j - i == L for i, j in pieces[:-1])
        assert eng.lib.tn_render_workspace_bytes(eng.rc, max(j - i for i, j in pieces)) <= (150 << 20) // streams  # a launch fits its slot's share
        _shards_against_the_frame(eng, o, d, n, 3)
        # the 1080p frame of config 4 under the default budget: 4 launches of 8 chunks, 2 slots
    eng = RayRenderEngine(model, chunk=CHUNK)
    assert eng.frame_launch_rays(1080 * 1920) == 8 * CHUNK and eng.frame_launch_rays(800 * 800) == 10 * CHUNK


def test_launch_forms_do_not_depend_on_the_previous_piece():
    """``_forms`` asked the library for a piece's kernel form with whatever ``rc.sample_split`` the PREVIOUS piece's
    launch had left behind — piece 0 of a frame was decided with 0, later pieces with the frame's split, and `sample_split == 1` moves
    the field's lane = ray threshold from 8 192 to 57 344 rays.  One stream, launches of 24 576 rays (between the two thresholds), a
    forced ``sample_split=1`` and the library's own choice: every full-size piece gets the same forms, a shard that STARTS in a later
    launch gets the form ``render`` used there, the shards equal the frame bit for bit, and ``config.sample_split`` reaches the engine."""
    from thermo_nerf_amd import synthetic
    from thermo_nerf_amd.engine import RayRenderEngine

    model, _, _ = _model()
    h, w = 250, 400  # 100 000 rays, chunk 8192: 13 chunks
    n = h * w
    o3, d3, _ = synthetic.orbit_camera_rays(h, w, view=5)
    o, d = o3.reshape(-1, 3).contiguous().to(DEV), d3.reshape(-1, 3).contiguous().to(DEV)
    eng = RayRenderEngine(model, chunk=8192, streams=1, max_workspace_bytes=45 << 20)
    L = eng.frame_launch_rays(n)
    assert 8192 <= L < 57344 and len(eng._launch_pieces(0, n, n)) >= 4
    _, _, fld = model._c_structs()
    for split in (1, None, 2):
        forms = [eng._forms(fld, n, i, split) for i, j in eng._launch_pieces(0, n, n) if j - i == L]
        assert len(set(forms)) == 1, (split, forms)
        eng.rc.sample_split = 7  # whatever a previous launch left behind must not matter ...
        assert eng._forms(fld, n, 0, split) == forms[0]
        assert eng.rc.sample_split == 7  # ... and is restored
        eng.rc.sample_split = 0
        _shards_against_the_frame(eng, o, d, n, 3, sample_split=split)
    assert eng._forms(fld, n, 0, 1)[2] == 1 and eng._forms(fld, n, 0, 1)[1] != eng._forms(fld, n, 0, None)[1]  # 24 576 rays: serial march -> one ray per wave
    # config.sample_split = 1 ("never": the serial march's bits) is the engine's default request, as it is model.get_outputs'
    want = {k: v.clone() for k, v in eng.render(o, d, sample_split=1).items()}
    model.config.sample_split = 1
    try:
        got = eng.render(o, d)
        torch.cuda.synchronize()
        assert all(torch.equal(got[k], want[k]) for k in want)
        assert eng._forms(fld, n, 0, None) == eng._forms(fld, n, 0, 1)
    finally:
        model.config.sample_split = 0


def test_more_ranks_than_tiles_and_planes_on_the_bundle():
    """(a) 130 rays over 8 ranks = 3 tiles: five ranks own EMPTY runs that start at the frame's end (130: not a multiple of 64) —
    render_shard returns empty outputs and neutral bounds instead of raising while its peers wait in the all-reduce.
    (b) per-ray near / far planes already on the bundle reach the shards as they reach the unsharded frame."""
    from thermo_nerf_amd import distributed as D
    from thermo_nerf_amd import synthetic
    from thermo_nerf_amd.engine import RayRenderEngine

    model, _, _ = _model()
    eng = RayRenderEngine(model, chunk=CHUNK)
    o3, d3, _ = synthetic.orbit_camera_rays(10, 13, view=1)
    o, d = o3.reshape(-1, 3).contiguous().to(DEV), d3.reshape(-1, 3).contiguous().to(DEV)
    blocks = [D.ray_block(130, r, 8) for r in range(8)]
    assert sum(1 for a, b in blocks if b == a) == 5 and blocks[-1] == (130, 130)
    out, bounds = eng.render_shard(o[130:], d[130:], 130, 130)
    assert out["rgb"].shape == (0, 3) and bool(torch.isinf(bounds).all())
    _shards_against_the_frame(eng, o, d, 130, 8)
    h, w = 120, 500
    n = h * w
    o3, d3, _ = synthetic.orbit_camera_rays(h, w, view=4)
    o, d = o3.reshape(-1, 3).contiguous().to(DEV), d3.reshape(-1, 3).contiguous().to(DEV)
    g = torch.Generator().manual_seed(3)
    nears = (0.05 + 0.2 * torch.rand(n, generator=g)).to(DEV)
    fars = (2.0 + 3.0 * torch.rand(n, generator=g)).to(DEV)
    with_planes = _shards_against_the_frame(eng, o, d, n, 5, nears, fars)
    plain = eng.render(o, d)
    assert not torch.equal(with_planes["depth"], plain["depth"])  # the planes were used
    with pytest.raises(ValueError):
        eng.render_shard(o[:64], d[:64], 0, n, nears=nears[:64])


def _fine_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from thermo_nerf_amd import distributed as D
        from thermo_nerf_amd.engine import RayRenderEngine

        model, _, _ = _model()
        o3, d3 = _frame_rays(55)
        eng = RayRenderEngine(model, chunk=CHUNK)
        got = D.render_frame_sharded_fine(eng, o3, d3, device=torch.device(DEV))  # default: the DEFAULT single-device frame's bits
        got_k = D.render_frame_sharded_fine(eng, o3, d3, device=torch.device(DEV), sample_split="shard")
        torch.cuda.synchronize()
        res = {}
        if rank == 0:
            flat = o3.reshape(-1, 3).contiguous(), d3.reshape(-1, 3).contiguous()
            want = eng.render(*flat)
            # (sample_split="shard": the segments per tile that suit one rank's run — a property of the frame at that world size)
            k = eng.shard_sample_split(D.ray_block(1080 * 1920, 0, world)[1])
            want_k = eng.render(*flat, sample_split=k)
            torch.cuda.synchronize()
            for key in D.OUTPUT_KEYS:
                res[key] = bool(torch.equal(got[key].reshape(want[key].shape), want[key]))
                res[key + "@shard_split"] = bool(torch.equal(got_k[key].reshape(want_k[key].shape), want_k[key]))
        dist.barrier()
        q.put((rank, res, None))
    except Exception:  # pragma: no cover - surfaced by the parent
        import traceback

        q.put((rank, {}, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_fine_sharded_frame_three_ranks_sharing_the_gpu():
    """render_frame_sharded_fine end to end: three gloo ranks on the one GPU render 691 200 rays each of a 1080p frame (10.5
    chunks: every rank boundary falls inside a chunk), exchange the chunk bounds with one all-reduce and all-gather the pixels;
    every rank's frame equals the unsharded one bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fine_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, r, err in res:
        assert err is None, f"rank {rank}:\n{err}"
    r0 = [r for rank, r, _ in res if rank == 0][0]
    assert r0 and all(r0.values()), r0


# ---- BASELINE config 5: per-GPU scene assignment = independent training replicas (no collective) ------------------------------------
def _train_five_steps(scene: int, dev=DEV):
    """Five optimisation steps (Adam, fused) on synthetic scene ``scene``: its own weights, rays, cameras and targets.  Returns
    {parameter name: CPU tensor} after the last step and the loss of every step."""
    import copy

    from tests import helpers
    from thermo_nerf_amd import synthetic
    from thermo_nerf_amd.rays import RayBundle

    cm, _, _ = helpers.build(("scene", "stress")[scene % 2], 48, camera_optimizer_mode="SO3xR3")
    gm = copy.deepcopy(cm).to(dev)
    gm.train()
    g = torch.Generator().manual_seed(1000 + scene)
    o, d, _ = synthetic.orbit_camera_rays(16, 16, view=2 + scene)
    o, d = o.reshape(-1, 3).contiguous().to(dev), d.reshape(-1, 3).contiguous().to(dev)
    R = o.shape[0]
    cam = torch.randint(0, 8, (R, 1), generator=g).to(dev)
    img, th = synthetic.analytic_scene(o.cpu(), d.cpu())
    batch = {"image": (img * (0.5 + 0.5 * scene)).clamp(0, 1).to(dev), "thermal": th.to(dev)}
    jit = torch.rand(5, 3, R, generator=g).to(dev)
    groups = gm.get_param_groups()
    opt = torch.optim.Adam([{"params": v} for v in groups.values()], lr=1e-2, eps=1e-15, fused=True)
    from thermo_nerf_amd import training as TR

    losses = []
    for i in range(5):
        gm.set_step(i)
        rb = gm.collider(RayBundle(origins=o, directions=d, camera_indices=cam))
        gm.camera_optimizer.apply_to_raybundle(rb)
        out = TR.get_outputs_train(gm, rb, jitter=jit[i].contiguous())
        loss = sum(gm.get_loss_dict(out, batch, gm.get_metrics_dict(out, batch)).values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return {n: p.detach().cpu().clone() for n, p in gm.named_parameters()}, losses


def _replica_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()  # both replicas train at the same time on the one GPU
        params, losses = _train_five_steps(rank)
        dist.barrier()
        q.put((rank, {k: v.numpy() for k, v in params.items()}, losses, None))
    except Exception:  # pragma: no cover - surfaced by the parent
        import traceback

        q.put((rank, {}, [], traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_config5_training_replicas_do_not_interact():
    """BASELINE config 5 (per-GPU scene assignment): N ranks train N different scenes with NO collective.  Two ranks (gloo)
    run five optimisation steps each, concurrently on the box's one GPU, on different scenes; each must end where a
    single-process run of its scene ends (nothing leaks between replicas through the engine table, the gradient arena, the
    per-device workspaces or stream state) and the two must differ from one another.  "Ends where" is not bit-for-bit: the
    table scatter's atomics make the last bits of a gradient sum order-dependent, and Adam (eps 1e-15) turns an entry whose
    gradient is rounding noise into a full +-lr step of either sign — two runs of ONE scene differ by ~5e-3 relative on the hash
    table after five steps, by ~4e-2 on the (initially zero) pose adjustments.  The losses agree to 1e-4, every tensor to
    0.1; the two scenes differ by O(1).)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    got = {}
    for rank, params, losses, err in res:
        assert err is None, f"rank {rank}:\n{err}"
        got[rank] = ({k: torch.from_numpy(v) for k, v in params.items()}, losses)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    for scene in (0, 1):
        want, want_losses = _train_five_steps(scene)
        params, losses = got[scene]
        assert len(losses) == 5 and all(abs(a - b) <= 1e-4 * abs(b) + 1e-7 for a, b in zip(losses, want_losses)), (losses, want_losses)
        for name, w in want.items():
            if w.numel() == 0:
                continue
            assert rel(params[name], w) <= 0.1, f"scene {scene} {name}: {rel(params[name], w):.2e}"
    moved = [n for n in got[0][0] if got[0][0][n].numel() and rel(got[0][0][n], got[1][0][n]) > 0.3]
    assert "camera_optimizer.pose_adjustment" in moved and "field.mlp_base.encoder.hash_table" in moved, moved


def test_bench_two_ranks_prints_one_parseable_line():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), on the gloo backend with
    both ranks on this box's one GPU (RCCL refuses two ranks per device; TN_BENCH_BACKEND is bench.py's switch for that): the LAST
    stdout line is the compact record — strict JSON below the limit — with the weak-scaling value, both strong-scaling frames
    (BASELINE config 4's 1080p x S=48 and the metric's 800x800 x S=192) and the transport record."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TN_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    last = lines[-1]
    assert len(last) <= 4096

    def no_constant(name):
        raise ValueError(name)

    line = json.loads(last, parse_constant=no_constant)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["unit"] == "rays/s" and line["value"] > 1e6 and line["dtype"] == "f32"
    assert line["value"] == pytest.approx(2 * 640000 / (line["ms_per_step"] * 1e-3), rel=1e-3)  # whole-job rays over the slowest rank's time
    assert line["rccl"]["backend"] == "gloo" and line["rccl"]["world_size"] == 2 and len(line["rccl"]["device_ids"]) == 2
    # (two ranks share this box's one GPU: a rank's proposal launch can sit behind the other rank's field launch, and then the
    # event-timed "dominant kernel" of the line is the proposal pass with its HBM roofline — one run in ten; on a GPU of its own
    # the field kernel dominates, tests/test_bench_line.py and the N=1 bench line)
    assert line["roofline"]["frac"] > 0 and line["roofline"]["bound"] in ("mfma", "hbm")
    for tag in ("strong_frame_1080p_S48", "strong_frame_800_S192"):
        v = line["variants"][tag]
        assert v["n_gpus"] == 2 and v["scaling"] == "strong" and v["value"] > 1e6 and v["ms_per_step"] > 0
    detail = json.loads(lines[-2])["bench_detail"]  # the full tree went out on the line before
    assert detail["variants"]["strong_frame_800_S192"]["config"]["rays_per_step"] == 640000


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """plain `python bench.py --gpus 2` — the command shape the driver uses, with NO launcher around it: bench.py starts the two ranks
    itself (torch.distributed.run, rendezvous on 127.0.0.1), rank 0 prints the one line with n_gpus 2, and ranks other than 0 got their
    weights through distributed.broadcast_model_ (the line says how many bytes).  gloo: both ranks share this box's one GPU.  And a
    --gpus that contradicts the launcher's WORLD_SIZE exits non-zero instead of measuring one GPU in silence."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-variants"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert line["n_gpus"] == 2 and line["rccl"]["world_size"] == 2 and line["rccl"]["backend"] == "gloo"
    assert line["value"] == pytest.approx(2 * 640000 / (line["ms_per_step"] * 1e-3), rel=1e-3)
    assert 70 < line["rccl"]["weights_broadcast_mb"] < 90  # main table 64 MiB + 2 x 5 MiB proposal tables + MLPs, embeddings

    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=root,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "contradicts" in (bad.stderr + bad.stdout)
