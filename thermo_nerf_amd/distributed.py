"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The hot path has no cross-ray term, so a frame shards by contiguous ROW BLOCKS (keeps the ray
coherence the field kernel's gathers rely on), weights are replicated (one broadcast at load, ~78 MB), and the only
exchange is ONE all-gather of the rendered pixels: 9 floats = 36 B per ray.  The single cross-ray quantity of the
reference — DepthRenderer("expected") clips to the [min, max] of the sample mid-points of its CALL, i.e. of one
``eval_num_rays_per_chunk`` chunk of the row-major frame — is reproduced exactly by sharding on CHUNK boundaries
(``render_frame_sharded(..., chunk=eval_num_rays_per_chunk)``): every rank renders whole chunks of the reference's own
chunking, so the sharded frame equals the single-device frame bit for bit, expected depth included, with no extra
collective.  (Row-block sharding, ``chunk=None``, clips per rank-local chunk instead; ``reduce_depth_bounds`` is the
two-scalar all-reduce for callers that want one frame-global clip.)

The reference itself only ever renders on one device [REF thermo_nerf/render/renderer.py:182-187]; nerfstudio's DDP is
training-only, so there is no NCCL call pattern to mirror here.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

OUTPUT_KEYS = ("rgb", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1", "thermal")
OUTPUT_WIDTHS = (3, 1, 1, 1, 1, 1, 1)


def broadcast_model_(model: torch.nn.Module, src: int = 0, group=None) -> int:
    """The one collective of model loading (weights replicated, "one RCCL broadcast of ~78 MB at load"): every
    parameter and buffer of ``model`` on every rank becomes rank ``src``'s, through ONE broadcast per dtype of a flat staging
    buffer (fp32: hash tables 64 + 2 x 5 MiB, MLPs, embeddings, pose adjustments) instead of one collective per tensor — over
    xGMI a broadcast is per-link bound, and ~40 small ones pay ~40 launch latencies.  Only rank ``src`` needs to have read the
    checkpoint.  Derived copies (dense re-layouts, MFMA blobs) are dropped so that the next forward rebuilds them from the
    received weights.  Returns the bytes broadcast.  The reference has no counterpart: it renders on one device
    [REF thermo_nerf/render/renderer.py:182-187] and trains through nerfstudio's DDP, whose constructor broadcasts module
    states the same way [REF thermo_nerf/nerfstudio_config/pipeline_tracking.py:26-27,44]."""
    tensors = [p.data for _, p in sorted(model.named_parameters(), key=lambda kv: kv[0])]
    tensors += [b for _, b in sorted(model.named_buffers(), key=lambda kv: kv[0]) if b is not None]
    total = 0
    by_kind: Dict = {}
    for t in tensors:
        if t.numel():
            by_kind.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, dev), group_t in sorted(by_kind.items(), key=lambda kv: (str(kv[0][0]), str(kv[0][1]))):
        flat = torch.empty(sum(t.numel() for t in group_t), dtype=dtype, device=dev)
        if dist.get_rank(group) == src:
            torch.cat([t.reshape(-1) for t in group_t], out=flat)
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in group_t:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        total += flat.numel() * flat.element_size()
    invalidate = getattr(model, "invalidate_prepared", None)
    if invalidate is not None:
        invalidate()
    return total


def row_block(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [start, end) of an image owned by ``rank``: contiguous blocks, sizes differ by at most one row."""
    base, extra = divmod(height, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_camera_rays(origins: Tensor, directions: Tensor, rank: int, world: int) -> Tuple[Tensor, Tensor, int, int]:
    """origins/directions [H,W,3] -> this rank's flat [rows*W,3] slices and its row range."""
    h = origins.shape[0]
    r0, r1 = row_block(h, rank, world)
    return (origins[r0:r1].reshape(-1, 3).contiguous(), directions[r0:r1].reshape(-1, 3).contiguous(), r0, r1)


def chunk_block(num_rays: int, chunk: int, rank: int, world: int) -> Tuple[int, int]:
    """Rays [start, end) of a row-major frame owned by ``rank`` when the frame is cut on the boundaries of the reference's
    chunks (``eval_num_rays_per_chunk`` rays each, REF config_thermal_nerf.py:30): contiguous runs of whole chunks, counts
    differing by at most one chunk; the frame's last (short) chunk belongs to the last rank that owns any."""
    n_chunks = (num_rays + chunk - 1) // chunk
    c0, c1 = row_block(n_chunks, rank, world)
    return min(c0 * chunk, num_rays), min(c1 * chunk, num_rays)


def pack_outputs(out: Dict[str, Tensor]) -> Tensor:
    """dict of [n,C] -> [n,9] (the 36 B/ray tuple that crosses xGMI)."""
    return torch.cat([out[k] for k in OUTPUT_KEYS], dim=1)


def unpack_outputs(packed: Tensor) -> Dict[str, Tensor]:
    res, c = {}, 0
    for k, w in zip(OUTPUT_KEYS, OUTPUT_WIDTHS):
        res[k] = packed[:, c:c + w]
        c += w
    return res


def gather_frame(local: Dict[str, Tensor], height: int, width: int, group=None, counts=None) -> Dict[str, Tensor]:
    """All-gather the per-rank ray ranges into full [H,W,C] images on every rank.  ``counts``: rays per rank in rank order
    (default: row blocks); uneven counts are supported."""
    world = dist.get_world_size(group)
    packed = pack_outputs(local).contiguous()
    if counts is None:
        counts = [(row_block(height, r, world)[1] - row_block(height, r, world)[0]) * width for r in range(world)]
    if len(set(counts)) == 1:
        full = torch.empty((sum(counts), packed.shape[1]), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(full, packed, group=group)
    else:
        # uneven row blocks: pad every shard to the largest one (collectives want equal sizes), trim after
        cmax = max(counts)
        padded = torch.zeros((cmax, packed.shape[1]), dtype=packed.dtype, device=packed.device)
        padded[: packed.shape[0]] = packed
        allp = torch.empty((world * cmax, packed.shape[1]), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(allp, padded, group=group)
        full = torch.cat([allp[r * cmax: r * cmax + counts[r]] for r in range(world)], dim=0)
    return {k: v.reshape(height, width, -1) for k, v in unpack_outputs(full).items()}


def reduce_depth_bounds(lo: Tensor, hi: Tensor, group=None) -> Tuple[Tensor, Tensor]:
    """Global [min, max] of the sample mid-points across ranks (two scalars)."""
    lo, hi = lo.clone(), hi.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return lo, hi


def render_frame_sharded(render_fn: Callable[[Tensor, Tensor], Dict[str, Tensor]], origins: Tensor, directions: Tensor,
                         group=None, device: Optional[torch.device] = None, chunk: Optional[int] = None) -> Dict[str, Tensor]:
    """Render one [H,W] camera ray bundle sharded over the process group; every rank returns the full [H,W,C] images.

    ``render_fn(origins[n,3], directions[n,3]) -> dict of [n,C]`` is the per-rank renderer (``RayRenderEngine.render`` on
    the GPU box).  ``chunk=None``: contiguous row blocks.  ``chunk=k``: contiguous runs of whole k-ray chunks of the
    row-major frame — with ``k = eval_num_rays_per_chunk`` and an engine of the same chunk size every rank renders exactly
    the calls the single-device loop makes [REF render/renderer.py:182-187], so the result is identical to it."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    h, w = origins.shape[:2]
    if chunk is None:
        o, d, _, _ = shard_camera_rays(origins, directions, rank, world)
        counts = None
    else:
        n = h * w
        r0, r1 = chunk_block(n, int(chunk), rank, world)
        o = origins.reshape(-1, 3)[r0:r1].contiguous()
        d = directions.reshape(-1, 3)[r0:r1].contiguous()
        counts = [chunk_block(n, int(chunk), r, world)[1] - chunk_block(n, int(chunk), r, world)[0] for r in range(world)]
    if device is not None:
        o, d = o.to(device), d.to(device)
    if o.shape[0] == 0:  # more ranks than chunks: this rank idles and only takes part in the gather
        dev = o.device
        local = {k: torch.empty((0, wd), dtype=torch.float32, device=dev) for k, wd in zip(OUTPUT_KEYS, OUTPUT_WIDTHS)}
    else:
        local = render_fn(o, d)
    return gather_frame(local, h, w, group=group, counts=counts)


def ray_block(num_rays: int, rank: int, world: int, align: int = 64) -> Tuple[int, int]:
    """Rays [start, end) of a row-major frame owned by ``rank`` when the frame is cut into ``world`` contiguous runs whose
    boundaries are multiples of ``align`` rays (64 = one wavefront tile of the lane = ray kernels): counts differ by at most one tile
    ``align``, whatever the reference's chunk size — 10 chunks over 8 ranks are 80 000 rays each, not 2, 2, 1, 1, 1, 1, 1, 1 chunks."""
    tiles = (num_rays + align - 1) // align
    t0, t1 = row_block(tiles, rank, world)
    return min(t0 * align, num_rays), min(t1 * align, num_rays)


def render_frame_sharded_fine(engine, origins: Tensor, directions: Tensor, group=None, device: Optional[torch.device] = None,
                              align: int = 64, nears: Optional[Tensor] = None, fars: Optional[Tensor] = None,
                              sample_split=None) -> Dict[str, Tensor]:
    """``render_frame_sharded`` with EVEN shards (``ray_block``) that still reproduces the single-device frame bit for bit (with the
    default ``sample_split``; see below).  The
    one cross-ray quantity — the expected-depth clip to the [min, max] sample mid-point of each ``engine.chunk``-ray chunk of
    the frame — is restored by exchanging the per-chunk bounds: a rank renders the pieces of the chunks its run overlaps
    (``engine.render_shard``), ONE all-reduce(min) of 2 floats per chunk of the frame (max as min of the negation) joins the
    bounds of chunks split between ranks, ``engine.apply_depth_bounds`` clips, and the pixels are all-gathered as before.
    ``nears`` / ``fars`` [H,W,1] (or [H*W]): per-ray planes the bundle already carries, sliced like the origins (absent: the
    engine's collider planes, as in ``RayRenderEngine.render``).
    ``sample_split``: segments per 64-ray tile of the field pass.  None (default): the unsharded frame's own choice, so the result
    is ``engine.render(frame)`` bit for bit whatever the number of ranks.  "shard": a PERFORMANCE opt-in — what suits the size of ONE
    RANK's run (``engine.shard_sample_split``: an 80 000-ray run is 1 250 tiles on 2 048 wave slots — marched whole it lasts as long
    as 2 048; in 8 segments each it lasts 5 short rounds), the same value on every rank; the frame then equals
    ``engine.render(frame, sample_split=k)`` bit for bit, and since k follows the WORLD SIZE the last bits of a pixel change with the
    number of ranks (another association of the same sum, within 3e-6 of the serial march).  k: forced.
    ``engine``: a RayRenderEngine (or anything with its ``render_shard`` / ``apply_depth_bounds``)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    h, w = origins.shape[:2]
    n = h * w
    r0, r1 = ray_block(n, rank, world, align)
    if sample_split == "shard":
        pick = getattr(engine, "shard_sample_split", None)
        sample_split = pick(ray_block(n, 0, world, align)[1]) if pick is not None else None
    o = origins.reshape(-1, 3)[r0:r1].contiguous()
    d = directions.reshape(-1, 3)[r0:r1].contiguous()
    if device is not None:
        o, d = o.to(device), d.to(device)
    planes = {}
    if nears is not None or fars is not None:
        if nears is None or fars is None:
            raise ValueError("pass both nears and fars, or neither")
        planes = {"nears": nears.reshape(-1)[r0:r1].contiguous().to(o.device), "fars": fars.reshape(-1)[r0:r1].contiguous().to(o.device)}
    if sample_split is not None:
        planes["sample_split"] = int(sample_split)
    local, bounds = engine.render_shard(o, d, r0, n, **planes)
    if world > 1:
        key = torch.stack([bounds[:, 0], -bounds[:, 1]], dim=1).contiguous()
        dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group)
        bounds = torch.stack([key[:, 0], -key[:, 1]], dim=1).contiguous()
    engine.apply_depth_bounds(local, r0, bounds)
    counts = [ray_block(n, r, world, align)[1] - ray_block(n, r, world, align)[0] for r in range(world)]
    return gather_frame(local, h, w, group=group, counts=counts)


class PipelinedFrameGather:
    """Double-buffered asynchronous all-gather of whole rendered frames (weak scaling: every rank renders its own frame).

    ``submit`` packs this rank's outputs into the next [n,9] buffer and launches the collective WITHOUT making the compute
    stream wait for it, so frame n+1 renders while frame n crosses xGMI (7 links x ~153 GB/s point-to-point: a 23 MB
    frame per rank is a few ms on a ring, against ~18 ms of rendering).  A buffer is reused only after the collective that
    read it has completed (``wait`` on its handle, which on RCCL orders streams and does not block the host)."""

    def __init__(self, num_rays: int, world: int, device, depth: int = 2, group=None) -> None:
        self.group, self.depth = group, depth
        self.packed = [torch.empty((num_rays, 9), dtype=torch.float32, device=device) for _ in range(depth)]
        self.gathered = [torch.empty((world * num_rays, 9), dtype=torch.float32, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.count = 0

    def submit(self, out: Dict[str, Tensor]) -> int:
        k = self.count % self.depth
        if self.work[k] is not None:
            self.work[k].wait()
        torch.cat([out[key] for key in OUTPUT_KEYS], dim=1, out=self.packed[k])
        self.work[k] = dist.all_gather_into_tensor(self.gathered[k], self.packed[k], group=self.group, async_op=True)
        self.count += 1
        return k

    def finish(self) -> None:
        for k, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[k] = None

    def frames(self, k: int, world: int) -> Tensor:
        """[world, n, 9] view of slot k (valid after finish() or after the slot's handle was waited on)."""
        return self.gathered[k].view(world, -1, 9)
