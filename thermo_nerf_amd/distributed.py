"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The hot path has no cross-ray term, so a frame shards by contiguous ROW BLOCKS (keeps the ray
coherence the field kernel's gathers rely on), weights are replicated (one broadcast at load, ~78 MB), and the only
exchange is ONE all-gather of the rendered pixels: 9 floats = 36 B per ray.  The single cross-ray quantity of the
reference — DepthRenderer("expected") clips to the call-global [min, max] of the sample mid-points — is reproduced by
an all-reduce(min/max) of two floats when ``exact_depth_clip`` is requested.

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


def pack_outputs(out: Dict[str, Tensor]) -> Tensor:
    """dict of [n,C] -> [n,9] (the 36 B/ray tuple that crosses xGMI)."""
    return torch.cat([out[k] for k in OUTPUT_KEYS], dim=1)


def unpack_outputs(packed: Tensor) -> Dict[str, Tensor]:
    res, c = {}, 0
    for k, w in zip(OUTPUT_KEYS, OUTPUT_WIDTHS):
        res[k] = packed[:, c:c + w]
        c += w
    return res


def gather_frame(local: Dict[str, Tensor], height: int, width: int, group=None) -> Dict[str, Tensor]:
    """All-gather the per-rank row blocks into full [H,W,C] images on every rank (uneven row counts supported)."""
    world = dist.get_world_size(group)
    packed = pack_outputs(local).contiguous()
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
                         group=None, device: Optional[torch.device] = None) -> Dict[str, Tensor]:
    """Render one [H,W] camera ray bundle with the rows sharded over the process group.

    ``render_fn(origins[n,3], directions[n,3]) -> dict of [n,C]`` is the per-rank renderer (RayRenderEngine.render on
    the GPU box).  Every rank returns the full [H,W,C] images."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    h, w = origins.shape[:2]
    o, d, _, _ = shard_camera_rays(origins, directions, rank, world)
    if device is not None:
        o, d = o.to(device), d.to(device)
    local = render_fn(o, d)
    return gather_frame(local, h, w, group=group)


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
