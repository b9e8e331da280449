"""Training step of the hot path on the HIP kernels.

The reference trains through torch autograd over nerfstudio's modules: ``ThermalNerfModel.get_outputs`` in train mode
[REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:210-275], ``get_loss_dict`` [REF :277-326] and nerfstudio's
``interlevel_loss`` / ``distortion_loss``.  Here the same graph is ONE ``torch.autograd.Function`` (``RenderTrain``)
whose forward chains the taped C-ABI stages of ``include/thermonerf_hip.h`` (hash encode -> Linear layers -> density
-> weights -> compositing, per level) and whose backward chains their adjoints; the two regularisers are Functions
of their own.  torch is used for what it is here for: the autograd tape between these Functions, the two MSE
reductions over [R,3]/[R,1] pixels, and the optimizer.

What carries gradient (as in nerfstudio): the final level through rgb / thermal / accumulation and its weights; the
proposal levels only through their weights (PDFSampler detaches the sample positions), and only on steps where
ProposalNetworkSampler's update schedule says so; the ray origins / directions through the sample positions and the
SH basis when a camera optimizer makes them depend on ``pose_adjustment``.  Not differentiated: median/expected depth.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _hip
from .rays import RayBundle
from .samplers import LazyRaySamples, draw_jitter, jitter_levels, linspace_bins, pdf_positions

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def _f32(shape, dev) -> Tensor:
    return torch.empty(shape, dtype=torch.float32, device=dev)


class _StreamOverride(threading.local):
    """per thread (autograd runs a device's backward on its own thread): [raw stream handle or None]"""

    def __init__(self) -> None:
        self.slot: List[Optional[int]] = [None]

    def __getitem__(self, k: int) -> Optional[int]:
        return self.slot[k]

    def __setitem__(self, k: int, v: Optional[int]) -> None:
        self.slot[k] = v


_STREAM_OVERRIDE = _StreamOverride()


def _stream():
    """the HIP stream the next launch goes to: torch's current stream, or the second stream of the step while its section of
    the backward is being queued (set explicitly: `with torch.cuda.stream(...)` costs ~0.1 ms of host time per use — its
    enter / exit look the current device up through torch.cuda.is_available() and os.environ)"""
    forced = _STREAM_OVERRIDE[0]
    return _hip.current_stream() if forced is None else forced


# --------------------------------------------------------------------------------------------------
# thin wrappers: one per C entry point (allocate the output, call, check)
# --------------------------------------------------------------------------------------------------
def hash_encode_fwd(grid, space, pos: Tensor) -> Tuple[Tensor, Tensor]:
    n = pos.shape[0]
    enc = _f32((n, 2 * grid.num_levels), pos.device)
    sel = _f32((n,), pos.device)
    _hip.check(_hip.load().tn_hash_encode_fwd(grid, space, pos.data_ptr(), n, enc.data_ptr(), sel.data_ptr(), _stream()),
               "tn_hash_encode_fwd")
    return enc, sel


_SPREAD_WS: Dict = {}


def _atomic_levels(lib, grid, space, pos: Tensor, d_enc: Tensor, d_table: Tensor, lo: int, hi: int, spread: bool) -> None:
    """levels [lo, hi) with global atomics; ``spread``: the coarsest levels through private dense copies (tn_hash_encode_bwd_spread)"""
    n = pos.shape[0]
    args = (grid, space, pos.data_ptr(), d_enc.data_ptr(), n, d_table.data_ptr(), lo, hi)
    need = lib.tn_hash_encode_bwd_spread_workspace_bytes(grid) if spread and lo == 0 else 0
    if need:
        key = (pos.device, _stream(), need)
        ws = _SPREAD_WS.get(key)
        if ws is None:
            ws = _SPREAD_WS[key] = torch.empty(need, dtype=torch.uint8, device=pos.device)
        _hip.check(lib.tn_hash_encode_bwd_spread(*args, ws.data_ptr(), need, _stream()), "tn_hash_encode_bwd_spread")
    else:
        _hip.check(lib.tn_hash_encode_bwd_levels(*args, _stream()), "tn_hash_encode_bwd_levels")


_SIDE_STREAMS: Dict = {}


def _step_streams(dev) -> Tuple["torch.cuda.Stream", "torch.cuda.Stream", "torch.cuda.Stream"]:
    """(torch's current stream on ``dev``, the second and the third stream of that stream's training step), as Stream objects —
    looked up by the raw handle, made once per (device, main stream)"""
    key = (dev, _hip.current_stream())
    trio = _SIDE_STREAMS.get(key)
    if trio is None:
        # (two streams on their own hardware queues: _hip.concurrent_streams — `torch.cuda.Stream()` twice may hand out a pair that
        # shares a queue with each other or with the main stream, and the step's overlap is gone)
        side = _hip.concurrent_streams(dev, 2)
        trio = _SIDE_STREAMS[key] = (torch.cuda.current_stream(dev), side[0], side[1])
    return trio


def hash_encode_bwd(grid, space, pos: Tensor, d_enc: Tensor, d_table: Tensor, bucketed=False, spread: bool = True,
                    overlap: bool = False, side_work=None, keep: Optional[list] = None, defer: bool = False) -> bool:
    """d_table += adjoint of the hash encoding.  ``bucketed=False``: the global-atomic scatter (tn_hash_encode_bwd).
    ``True`` (config.bucketed_table_scatter): from the level the library names (scaling >= 200, enough table slices: the
    field's grid, not the proposal grids) the contributions are written out as records bucketed by the owning table slice and
    summed in LDS (tn_hash_encode_bwd_sorted: no global atomics; 25 B of scratch per (sample, level, corner pair)), the
    coarser levels keep the atomics.  An int (tests): bucketed from that level on, whatever the library advises.
    ``spread`` (config.spread_coarse_scatter): the coarsest levels of the atomic part accumulate in private dense copies.
    ``overlap`` (config.overlap_table_scatter): the two parts write disjoint levels of d_table and wait on different units (the
    memory-side atomic unit / LDS + streaming), so the bucketed part runs on a second stream beside the atomic part; the
    calling stream continues when both are done.  ``side_work``: a callable with more launches that depend on neither part
    (the step's ray-level adjoints): queued on the calling stream behind the atomic part, before the join — the bucketed part
    on the second stream is the longer of the two.  ``keep``: the caller queues this call on another stream than the
    allocator's (_STREAM_OVERRIDE) and holds the temporaries appended here — the record workspace — until it has joined that
    stream (freed on return, the block could be handed to the main stream while the side stream still works in it).
    ``defer`` (config.deferred_table_update, with ``overlap`` and a bucketed part): BOTH parts go to side streams — the bucketed one
    to the second, the atomic one to the third — and the calling stream does not wait for them: it runs ``side_work`` and returns
    True; the join is the caller's business (_hip.defer / _hip.join_pending hold the temporaries until then).  Returns False when
    the scatter was joined as usual."""
    if side_work is not None and not overlap:
        hash_encode_bwd(grid, space, pos, d_enc, d_table, bucketed, spread, keep=keep)
        side_work()
        return False
    lib = _hip.load()
    n = pos.shape[0]
    first = -1
    if bucketed is True:
        first = lib.tn_hash_encode_bwd_sorted_first_level(grid, n)
    elif bucketed is not False and bucketed is not None:
        first = int(bucketed)
    need = lib.tn_hash_encode_bwd_sorted_workspace_bytes(grid, n, first) if first >= 0 else 0
    if not need:
        _atomic_levels(lib, grid, space, pos, d_enc, d_table, 0, grid.num_levels, spread)
        if side_work is not None:
            side_work()
        return False
    try:
        ws = torch.empty(need, dtype=torch.uint8, device=pos.device)  # 25 B per (sample, level, corner pair): 0.14-0.6 GB, from torch's caching allocator
        if keep is not None:
            keep.append(ws)
    except torch.cuda.OutOfMemoryError:  # no room for the records: the same sums through the global atomics
        _atomic_levels(lib, grid, space, pos, d_enc, d_table, 0, first, spread)
        _hip.check(lib.tn_hash_encode_bwd_levels(grid, space, pos.data_ptr(), d_enc.data_ptr(), n, d_table.data_ptr(), first,
                                                 grid.num_levels, _stream()), "tn_hash_encode_bwd_levels")
        if side_work is not None:
            side_work()
        return False
    if overlap and first > 0 and defer:
        main, side, third = _step_streams(pos.device)
        side.wait_stream(main)  # d_enc, positions and the cleared d_table are the main stream's work so far
        third.wait_stream(main)
        _hip.check(lib.tn_hash_encode_bwd_sorted(grid, space, pos.data_ptr(), d_enc.data_ptr(), n, d_table.data_ptr(), first,
                                                 ws.data_ptr(), need, side.cuda_stream), "tn_hash_encode_bwd_sorted")
        saved = _STREAM_OVERRIDE[0]
        _STREAM_OVERRIDE[0] = third.cuda_stream  # (the spread copies' workspace is keyed by the stream: the third's own)
        try:
            _atomic_levels(lib, grid, space, pos, d_enc, d_table, 0, first, spread)
        finally:
            _STREAM_OVERRIDE[0] = saved
        if side_work is not None:
            side_work()  # on the calling stream, beside both parts
        # (NOT d_table itself: autograd's AccumulateGrad takes a returned gradient over without a copy only while nobody else
        # holds that tensor object — a second reference makes it CLONE the table gradient on the calling stream, i.e. before the
        # side streams have written it.  The caller keeps the storage alive through another view: the arena's flat buffer.)
        _hip.defer(pos.device, [side, third], [ws, pos, d_enc])
        return True
    if overlap and first > 0:
        main, side, _ = _step_streams(pos.device)
        side.wait_stream(main)  # d_enc, positions and the cleared d_table are the main stream's work so far
        _hip.check(lib.tn_hash_encode_bwd_sorted(grid, space, pos.data_ptr(), d_enc.data_ptr(), n, d_table.data_ptr(), first,
                                                 ws.data_ptr(), need, side.cuda_stream), "tn_hash_encode_bwd_sorted")
        _atomic_levels(lib, grid, space, pos, d_enc, d_table, 0, first, spread)
        if side_work is not None:
            side_work()  # behind the atomic part: the bucketed part is the longer of the two (timeline in DESIGN 5.6)
        main.wait_stream(side)  # also what keeps `ws`, d_enc and pos (main-stream allocations) from being reused too early
        return False
    if first > 0:
        _atomic_levels(lib, grid, space, pos, d_enc, d_table, 0, first, spread)
    _hip.check(lib.tn_hash_encode_bwd_sorted(grid, space, pos.data_ptr(), d_enc.data_ptr(), n, d_table.data_ptr(), first, ws.data_ptr(),
                                             need, _stream()), "tn_hash_encode_bwd_sorted")
    if side_work is not None:
        side_work()
    return False


def linear_fwd(x: Tensor, x_off: int, ldx: int, lin, act: int, n: int) -> Tensor:
    y = _f32((n, lin.out_dim), x.device)
    _hip.check(_hip.load().tn_linear_fwd(x.data_ptr() + 4 * x_off, ldx, lin, act, n, y.data_ptr(), lin.out_dim, _stream()),
               "tn_linear_fwd")
    return y


_LIN_WS: Dict = {}


def _linear_workspace(dev) -> Tensor:
    """per-device scratch for tn_linear_bwd's partial weight gradients (calls on one stream are ordered, so one buffer)"""
    key = (dev, _stream())
    ws = _LIN_WS.get(key)
    if ws is None:
        ws = torch.empty(_hip.load().tn_linear_bwd_workspace_bytes(), dtype=torch.uint8, device=dev)
        _LIN_WS[key] = ws
    return ws


def linear_bwd(x: Tensor, x_off: int, ldx: int, y: Optional[Tensor], dy: Tensor, ldy: int, lin, act: int, n: int,
               dx: Optional[Tensor], dx_off: int, lddx: int, accumulate: bool, d_w: Optional[Tensor],
               d_b: Optional[Tensor]) -> None:
    ws = _linear_workspace(dy.device)
    _hip.check(_hip.load().tn_linear_bwd(
        x.data_ptr() + 4 * x_off, ldx, None if y is None else y.data_ptr(), dy.data_ptr(), ldy, lin, act, n,
        None if dx is None else dx.data_ptr() + 4 * dx_off, lddx, 1 if accumulate else 0,
        None if d_w is None else d_w.data_ptr(), None if d_b is None else d_b.data_ptr(), ws.data_ptr(), ws.numel(),
        _stream()), "tn_linear_bwd")


_CHAIN_WS: Dict = {}
_FUSED_WS: Dict = {}
_ZEROS: Dict = {}


def _fused_bwd_workspace(dev, R: int, S: int) -> Tensor:
    """per-device workspace of tn_field_bwd_fused (a slab of parameter gradients per persistent block and launch + the heads'
    adjoints of mlp_base's outputs), grown to the largest batch seen"""
    need = _hip.load().tn_field_bwd_fused_workspace_bytes(R, S)
    key = (dev, _stream())
    ws = _FUSED_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _FUSED_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    return ws


def _zeros_like_cached(dev, n: int) -> Tensor:
    """a read-only run of n zero floats on ``dev`` (the geo rows of the ray-level mlp_head input)"""
    z = _ZEROS.get(dev)
    if z is None or z.numel() < n:
        z = _ZEROS[dev] = torch.zeros((max(n, 1 << 16),), dtype=torch.float32, device=dev)
    return z


def linear_chain_bwd(layers, y_top: Optional[Tensor], act_top: int, dy: Tensor, lddy: int, n: int, dx: Optional[Tensor],
                     dx_off: int, lddx: int, accumulate: bool) -> None:
    """Backward of consecutive Linear(+activation) layers in one launch.  ``layers``: top (nearest the loss) first, each
    (tn_linear, x tensor, x column offset, ldx, activation that produced x, d_weight, d_bias)."""
    lib = _hip.load()
    dev = dy.device
    key = (dev, _stream())  # calls on one stream are ordered; two streams must not share the partial-sum slabs
    ws = _CHAIN_WS.get(key)
    if ws is None:
        ws = _CHAIN_WS[key] = torch.empty(lib.tn_linear_chain_bwd_workspace_bytes(), dtype=torch.uint8, device=dev)
    arr = (_hip.tn_chain_layer * len(layers))()
    for k, (lin, x, x_off, ldx, act_x, d_w, d_b) in enumerate(layers):
        arr[k].lin = lin
        arr[k].x = x.data_ptr() + 4 * x_off
        arr[k].ldx, arr[k].act_x = ldx, act_x
        arr[k].d_weight = None if d_w is None else d_w.data_ptr()
        arr[k].d_bias = None if d_b is None else d_b.data_ptr()
    _hip.check(lib.tn_linear_chain_bwd(arr, len(layers), None if y_top is None else y_top.data_ptr(), act_top, dy.data_ptr(), lddy,
                                       n, None if dx is None else dx.data_ptr() + 4 * dx_off, lddx, 1 if accumulate else 0,
                                       ws.data_ptr(), ws.numel(), _stream()), "tn_linear_chain_bwd")


def weights_fwd(deltas: Tensor, density: Tensor) -> Tensor:
    R, n = deltas.shape
    w = _f32((R, n), deltas.device)
    _hip.check(_hip.load().tn_weights_fwd(deltas.data_ptr(), density.data_ptr(), R, n, w.data_ptr(), _stream()),
               "tn_weights_fwd")
    return w


def weights_bwd(deltas: Tensor, density: Tensor, g_w: Tensor) -> Tensor:
    R, n = deltas.shape
    g = _f32((R, n), deltas.device)
    _hip.check(_hip.load().tn_weights_bwd(deltas.data_ptr(), density.data_ptr(), g_w.data_ptr(), R, n, g.data_ptr(),
                                          _stream()), "tn_weights_bwd")
    return g


class _LevelTape:
    """What one sampling level keeps for its backward."""

    __slots__ = ("pos", "enc", "sel", "hid", "raw", "density", "deltas", "weights", "spacing", "eucl", "starts", "ends")


def _starts_ends(t: "_LevelTape") -> Tuple[Tensor, Tensor]:
    """contiguous [R,n] copies of the two edge columns of eucl [R,n+1], made once per level and step"""
    if getattr(t, "starts", None) is None:
        t.starts, t.ends = t.eucl[:, :-1].contiguous(), t.eucl[:, 1:].contiguous()
    return t.starts, t.ends


class _GradArena:
    """Zero-initialised gradient buffers of one backward pass as views of ONE flat allocation cleared by ONE fill (the
    adjoint kernels accumulate with += / atomics, so every parameter gradient starts at zero; a torch.zeros_like per
    parameter was ~50 fill launches per step).  A fresh arena per backward: the views become the parameters' .grad and must
    not alias the next step's."""

    def __init__(self, like: Dict[str, Tensor], dev, extra: int = 0, zero: bool = True) -> None:
        """``zero=False``: the flat buffer is allocated but not cleared — tn_train_step_fwd clears it on the step's second stream
        beside the forward (the 75 MB fill leaves the calling stream's critical path)"""
        self.like, self.dev = like, dev
        self.offsets: Dict[str, int] = {}
        total = 0
        for name, p in like.items():
            self.offsets[name] = total
            total += (p.numel() + 63) // 64 * 64  # 256-byte aligned views
        self._extra = total  # `extra` more zero floats for the step's other zero-initialised buffers (ray gradients, per-ray sums)
        self.flat = (torch.zeros if zero else torch.empty)((total + extra,), dtype=torch.float32, device=dev)

    def zeros(self, shape) -> Tensor:
        """a zero tensor carved from the arena's tail (falls back to a fresh allocation when the tail is used up)"""
        n = 1
        for k in shape:
            n *= k
        n_al = (n + 63) // 64 * 64
        if self._extra + n_al > self.flat.numel():
            return torch.zeros(shape, dtype=torch.float32, device=self.dev)
        out = self.flat[self._extra:self._extra + n].view(shape)
        self._extra += n_al
        return out

    def get(self, name: str) -> Tensor:
        p = self.like[name]
        off = self.offsets[name]
        return self.flat[off:off + p.numel()].view(p.shape)


def _frustum_positions(o: Tensor, d: Tensor, t: "_LevelTape") -> Tuple[Tensor, Tensor]:
    R, n1 = t.eucl.shape
    n = n1 - 1
    pos = _f32((R * n, 3), o.device)
    t.starts, t.ends, deltas = _f32((R, n), o.device), _f32((R, n), o.device), _f32((R, n), o.device)
    eucl = t.eucl if t.eucl.is_contiguous() else t.eucl.contiguous()
    _hip.check(_hip.load().tn_frustum_from_edges(o.data_ptr(), d.data_ptr(), eucl.data_ptr(), R, n, pos.data_ptr(),
                                                 t.starts.data_ptr(), t.ends.data_ptr(), deltas.data_ptr(), _stream()),
               "tn_frustum_from_edges")
    return pos, deltas


def _proposal_level_fwd(net_struct, o: Tensor, d: Tensor, spacing: Tensor, eucl: Tensor, fused: bool = True) -> _LevelTape:
    lib = _hip.load()
    t = _LevelTape()
    t.spacing, t.eucl = spacing, eucl
    t.pos, t.deltas = _frustum_positions(o, d, t)
    n = t.pos.shape[0]
    t.hid = None
    if fused:
        # encode + Linear(10,16)+ReLU + Linear(16,1) + trunc_exp in one launch; the hidden layer is not kept (the backward recomputes it)
        E = 2 * net_struct.grid.num_levels
        t.enc, t.sel, t.raw, t.density = _f32((n, E), o.device), _f32((n,), o.device), _f32((n, 1), o.device), _f32((n,), o.device)
        code = lib.tn_density_fwd_train(net_struct, t.pos.data_ptr(), n, t.enc.data_ptr(), t.sel.data_ptr(), t.raw.data_ptr(),
                                        t.density.data_ptr(), _stream())
        if code != -3:  # TN_ERR_UNSUPPORTED: a geometry other than the reference's 5 levels x hidden 16 -> the stage chain below
            _hip.check(code, "tn_density_fwd_train")
            t.weights = weights_fwd(t.deltas, t.density.view(t.deltas.shape))
            return t
    t.enc, t.sel = hash_encode_fwd(net_struct.grid, net_struct.space, t.pos)
    t.hid = linear_fwd(t.enc, 0, t.enc.shape[1], net_struct.l0, ACT_RELU, n)
    t.raw = linear_fwd(t.hid, 0, t.hid.shape[1], net_struct.l1, ACT_NONE, n)
    t.density = _f32((n,), o.device)
    _hip.check(lib.tn_density_act_fwd(t.raw.data_ptr(), 1, t.sel.data_ptr(), net_struct.average_init_density, n,
                                      t.density.data_ptr(), _stream()), "tn_density_act_fwd")
    t.weights = weights_fwd(t.deltas, t.density.view(t.deltas.shape))
    return t


def _ray_grads_from_enc(grid, space, t: _LevelTape, g_enc: Tensor, g_o: Tensor, g_d: Tensor, keep: Optional[list] = None) -> None:
    """d loss / d (origins, directions) through the sample positions of one level (camera-pose optimisation)."""
    lib = _hip.load()
    n = t.pos.shape[0]
    R, per = t.deltas.shape
    g_pos = _f32((n, 3), g_enc.device)
    if keep is not None:
        keep.append(g_pos)
    _hip.check(lib.tn_hash_encode_bwd_input(grid, space, t.pos.data_ptr(), g_enc.data_ptr(), n, g_pos.data_ptr(), _stream()),
               "tn_hash_encode_bwd_input")
    starts, ends = _starts_ends(t)
    _hip.check(lib.tn_frustum_positions_bwd(g_pos.data_ptr(), starts.data_ptr(), ends.data_ptr(), R, per, g_o.data_ptr(),
                                            g_d.data_ptr(), _stream()), "tn_frustum_positions_bwd")


def _proposal_level_bwd(net_struct, t: _LevelTape, g_w: Tensor, grads: Dict[str, Tensor], prefix: str, like: Dict,
                        ray_grads: Optional[Tuple[Tensor, Tensor]] = None, chained: bool = True, bucketed: bool = False,
                        exp_min: float = -15.0, spread: bool = True, keep: Optional[list] = None) -> None:
    """``keep``: a list the temporaries are appended to — the caller queues this level on another stream than the allocator's
    and holds them until that stream has been joined (the tape-free / fused form only: no torch op in between)."""
    lib = _hip.load()
    n = t.pos.shape[0]
    g_density = weights_bwd(t.deltas, t.density.view(t.deltas.shape), g_w)
    names = [f"{prefix}.mlp_base.encoder.hash_table", f"{prefix}.mlp_base.mlp.layers.0.weight",
             f"{prefix}.mlp_base.mlp.layers.0.bias", f"{prefix}.mlp_base.mlp.layers.1.weight",
             f"{prefix}.mlp_base.mlp.layers.1.bias"]
    for k in names:
        if k not in grads:
            grads[k] = like.get(k)  # `like` is the step's _GradArena: zero-filled views
    E = t.enc.shape[1]
    g_enc = _f32((n, E), g_w.device)
    if keep is not None:
        keep += [g_density, g_enc, g_w]
    if t.hid is None:
        # the fused forward's counterpart: trunc_exp backward + both Linear layers' adjoints in one launch, hidden layer recomputed
        _hip.check(lib.tn_density_bwd_train(net_struct, t.enc.data_ptr(), t.raw.data_ptr(), t.sel.data_ptr(), g_density.data_ptr(), n,
                                            exp_min, g_enc.data_ptr(), grads[names[1]].data_ptr(), grads[names[2]].data_ptr(),
                                            grads[names[3]].data_ptr(), grads[names[4]].data_ptr(), _stream()), "tn_density_bwd_train")
    else:
        g_raw = _f32((n, 1), g_w.device)
        _hip.check(lib.tn_density_act_bwd(t.raw.data_ptr(), 1, t.sel.data_ptr(), net_struct.average_init_density, exp_min,
                                          g_density.data_ptr(), n, g_raw.data_ptr(), 1, 0, _stream()), "tn_density_act_bwd")
        H = t.hid.shape[1]
        if chained:
            linear_chain_bwd([(net_struct.l1, t.hid, 0, H, ACT_RELU, grads[names[3]], grads[names[4]]),
                              (net_struct.l0, t.enc, 0, E, ACT_NONE, grads[names[1]], grads[names[2]])],
                             None, ACT_NONE, g_raw, 1, n, g_enc, 0, E, False)
        else:
            g_hid = _f32((n, H), g_w.device)
            linear_bwd(t.hid, 0, H, None, g_raw, 1, net_struct.l1, ACT_NONE, n, g_hid, 0, H, False, grads[names[3]], grads[names[4]])
            linear_bwd(t.enc, 0, E, t.hid, g_hid, H, net_struct.l0, ACT_RELU, n, g_enc, 0, E, False, grads[names[1]], grads[names[2]])
    hash_encode_bwd(net_struct.grid, net_struct.space, t.pos, g_enc, grads[names[0]], bucketed, spread, keep=keep)
    if ray_grads is not None:
        _ray_grads_from_enc(net_struct.grid, net_struct.space, t, g_enc, *ray_grads, keep=keep)


class RenderTrain(torch.autograd.Function):
    """Train-mode ``get_outputs`` with a tape.  ``apply(model, origins, directions, nears, fars, camera_indices,
    jitter, updated, *parameters)``; returns (rgb [R,3], thermal [R,1], accumulation [R,1], w0 [R,P0,1], w1 [R,P1,1],
    w2 [R,S,1], depth, expected_depth, prop_depth_0, prop_depth_1, sp0, sp1, sp2, eu0, eu1, eu2)."""

    @staticmethod
    def forward(ctx, model, o: Tensor, d: Tensor, nears: Tensor, fars: Tensor, cam: Tensor, jitter: Tensor,
                updated: bool, *params: Tensor):
        lib = _hip.load()
        ctx.set_materialize_grads(False)  # outputs no loss touches arrive as None: their branches are skipped
        cfg = model.config
        dev = o.device
        R = o.shape[0]
        P = tuple(cfg.num_proposal_samples_per_ray)
        S = cfg.num_nerf_samples_per_ray
        # use_same_proposal_network: ONE HashMLPDensityField serves both levels [REF thermal_nerf_model.py:127-139]
        nets = len(model.proposal_networks)
        prop_structs = [model.proposal_networks[min(i, nets - 1)].train_struct() for i in range(2)]
        fld = model.field.train_struct()
        anneal = float(model.proposal_sampler._anneal)
        uniform = int(model.proposal_sampler.initial_sampler.uniform_spacing)  # REF thermal_nerf_model.py:164-170
        single = bool(cfg.use_single_jitter)  # REF thermal_nerf_model.py:176; False: one draw per bin edge
        jitter, jit = jitter_levels(jitter, R, (P[0], P[1], S), single)
        spacing_flags = uniform | (0 if single else 2)

        # ---- proposal levels ---------------------------------------------------------------------------------
        tapes: List[_LevelTape] = []
        prop_depths: List[Tensor] = []
        if updated:
            # the proposal networks take gradient this step: sample -> taped density -> weights, level by level
            spacing = _f32((R, P[0] + 1), dev)
            eucl = _f32((R, P[0] + 1), dev)
            _hip.check(lib.tn_sample_initial(linspace_bins(P[0], dev).data_ptr(), jit[0].data_ptr(), nears.data_ptr(),
                                             fars.data_ptr(), R, P[0], spacing_flags, spacing.data_ptr(), eucl.data_ptr(),
                                             _stream()),
                       "tn_sample_initial")
            counts = (P[1], S)
            for lvl in range(2):
                t = _proposal_level_fwd(prop_structs[lvl], o, d, spacing, eucl, bool(getattr(cfg, "fused_proposal_training", True)))
                tapes.append(t)
                n_out = counts[lvl]
                w_in = t.weights if anneal == 1.0 else torch.pow(t.weights, anneal)
                spacing, eucl = _f32((R, n_out + 1), dev), _f32((R, n_out + 1), dev)
                _hip.check(lib.tn_sample_pdf(w_in.data_ptr(), t.spacing.data_ptr(), pdf_positions(n_out + 1, dev, True).data_ptr(),
                                             jit[lvl + 1].data_ptr(), nears.data_ptr(), fars.data_ptr(), R, P[lvl], n_out,
                                             spacing_flags, spacing.data_ptr(), eucl.data_ptr(), _stream()), "tn_sample_pdf")
        else:
            # nerfstudio evaluates the proposal densities under no_grad on these steps (5 of 6 after warm-up): no tape is
            # needed, so both levels run as ONE fused kernel (tn_proposal_sample_fwd, train-mode semantics)
            if _step_call_applies(model, cfg):
                # ... and the whole forward chain of such a step as ONE C-ABI call (tn_train_step_fwd, round 6)
                return _StepCall.forward(ctx, model, o, d, nears, fars, cam, jitter, params, prop_structs, anneal, uniform, single)
            rc = _hip.tn_render_config()
            rc.num_proposal_samples[0], rc.num_proposal_samples[1], rc.num_nerf_samples = P[0], P[1], S
            rc.training, rc.pdf_anneal, rc.early_stop_transmittance, rc.kernel_family = 1, anneal, 0.0, 0
            rc.initial_sampler = uniform
            rc.per_sample_jitter = 0 if single else 1
            ins = _hip.tn_render_inputs()
            ins.origins, ins.directions, ins.nears, ins.fars = o.data_ptr(), d.data_ptr(), nears.data_ptr(), fars.data_ptr()
            ins.camera_indices, ins.jitter = cam.data_ptr(), jitter.data_ptr()
            ins.lin_bins0 = linspace_bins(P[0], dev).data_ptr()
            ins.u1 = pdf_positions(P[1] + 1, dev, True).data_ptr()
            ins.u2 = pdf_positions(S + 1, dev, True).data_ptr()
            outs = _hip.tn_render_outputs()
            ns = (P[0], P[1], S)
            sp = [_f32((R, k + 1), dev) for k in ns]
            eu = [_f32((R, k + 1), dev) for k in ns]
            ws_ = [_f32((R, k), dev) for k in ns[:2]]
            prop_depths = [_f32((R, 1), dev), _f32((R, 1), dev)]
            for i in range(3):
                outs.spacing_bins[i], outs.eucl_bins[i] = sp[i].data_ptr(), eu[i].data_ptr()
            outs.weights[0], outs.weights[1] = ws_[0].data_ptr(), ws_[1].data_ptr()
            outs.prop_depth_0, outs.prop_depth_1 = prop_depths[0].data_ptr(), prop_depths[1].data_ptr()
            need = lib.tn_render_workspace_bytes(rc, R)
            wsb = torch.empty(need, dtype=torch.uint8, device=dev)
            _hip.check(lib.tn_proposal_sample_fwd(prop_structs[0], prop_structs[1], rc, ins, outs, R, wsb.data_ptr(), need,
                                                  _stream()), "tn_proposal_sample_fwd")
            for i in range(2):
                t = _LevelTape()
                t.spacing, t.eucl, t.weights = sp[i], eu[i], ws_[i]
                tapes.append(t)
            spacing, eucl = sp[2], eu[2]

        # ---- final level: taped field ----------------------------------------------------------------------
        f = _LevelTape()
        f.spacing, f.eucl = spacing, eucl
        f.pos, f.deltas = _frustum_positions(o, d, f)
        N = R * S
        fused = None
        tape_free = bool(getattr(cfg, "tape_free_training", True))
        if cfg.fused_train_forward or tape_free:
            # MFMA fragments of the CURRENT weights (rebuilt per step); None: geometry the MFMA chain does not cover -> the
            # stage-by-stage entry points below
            fused = model.field.train_struct(prepare=True)
        tape_free = tape_free and fused is not None
        # config.deferred_table_update: the previous step's table scatter + Adam may still be running on the side streams; everything
        # above (camera optimizer, proposal pass, level geometry, field_prepare) needed neither — the field's table reads do
        _hip.join_pending(dev)
        h1 = bo = cin = c1 = c2 = t1 = t2 = None
        if tape_free:
            # no activation tape: the per-ray constant inputs of mlp_head.0 (SH(direction), appearance embedding) become a
            # per-ray bias [R,64]; the forward keeps enc / selector / density / rgb / thermal, tn_field_bwd_fused recomputes the rest
            ray_bias = _f32((R, 64), dev)
            _hip.check(lib.tn_ray_head_fwd(fld, d.data_ptr(), cam.data_ptr(), R, ray_bias.data_ptr(), _stream()), "tn_ray_head_fwd")
            f.enc, f.sel, f.density = _f32(((N + 63) // 64 * 64, 32), dev), _f32((N,), dev), _f32((N,), dev)  # enc: 64-sample tiles
            rgb_s, th_s = _f32((N, 3), dev), _f32((N, 1), dev)
            # (round 5) mlp_base's 16 output rows too, 64 B per sample: the backward's two head launches read them instead of
            # recomputing mlp_base (the split form only: the one-launch form needs the hidden layer anyway)
            keep_base = bool(getattr(cfg, "store_base_output", True)) and bool(getattr(cfg, "fused_backward_split", True))
            base_out = _f32((N, 16), dev) if keep_base else None
            # (round 5) camera-pose optimisation: d hash features / d position while the corner values are in registers (384 B per
            # sample) — the backward's position gradient then reads no table
            keep_jac = (o.requires_grad or d.requires_grad) and bool(getattr(cfg, "store_position_jacobian", True)) \
                and bool(getattr(cfg, "fused_backward_split", True))
            jac = _f32(((N + 63) // 64 * 64, 96), dev) if keep_jac else None
            _hip.check(lib.tn_field_fwd_train(fused, f.pos.data_ptr(), ray_bias.data_ptr(), R, S, f.enc.data_ptr(), f.sel.data_ptr(),
                                              f.density.data_ptr(), rgb_s.data_ptr(), th_s.data_ptr(), _hip.ptr(base_out),
                                              _hip.ptr(jac), _stream()),
                       "tn_field_fwd_train")
            bo = ray_bias  # (slot reuse in ctx.acts: the tape-free backward reads (ray_bias, rgb_s) ...
            h1 = base_out  # ... mlp_base's output rows, if kept ...
            cin = jac      # ... and the position Jacobian, if kept)
        elif fused is not None and cfg.fused_train_forward:
            # the whole field forward of the level in one launch; every tensor of the tape in the layout the adjoints read
            f.enc, f.sel, f.density = _f32((N, 32), dev), _f32((N,), dev), _f32((N,), dev)
            h1, bo = _f32((N, 64), dev), _f32((N, 16), dev)
            c1, c2, rgb_s = _f32((N, 64), dev), _f32((N, 64), dev), _f32((N, 3), dev)
            t1, t2, th_s = _f32((N, 64), dev), _f32((N, 64), dev), _f32((N, 1), dev)
            _hip.check(lib.tn_field_fwd_taped(fused, f.pos.data_ptr(), d.data_ptr(), cam.data_ptr(), R, S, f.enc.data_ptr(),
                                              f.sel.data_ptr(), h1.data_ptr(), bo.data_ptr(), f.density.data_ptr(),
                                              c1.data_ptr(), c2.data_ptr(), rgb_s.data_ptr(), t1.data_ptr(), t2.data_ptr(),
                                              th_s.data_ptr(), _stream()), "tn_field_fwd_taped")
            ldb = bo.shape[1]
            cin = _f32((N, 64), dev)  # the colour layer's input rows, which its weight gradient multiplies
            _hip.check(lib.tn_color_input_fwd(fld, d.data_ptr(), bo.data_ptr() + 4, ldb, cam.data_ptr(), 1, R, S,
                                              cin.data_ptr(), _stream()), "tn_color_input_fwd")
        else:
            f.enc, f.sel = hash_encode_fwd(fld.grid, fld.space, f.pos)
            E = f.enc.shape[1]
            h1 = linear_fwd(f.enc, 0, E, fld.base0, ACT_RELU, N)
            bo = linear_fwd(h1, 0, h1.shape[1], fld.base1, ACT_NONE, N)  # [N, 1 + geo]: raw density | geo features
            ldb = bo.shape[1]
            f.density = _f32((N,), dev)
            _hip.check(lib.tn_density_act_fwd(bo.data_ptr(), ldb, f.sel.data_ptr(), fld.average_init_density, N,
                                              f.density.data_ptr(), _stream()), "tn_density_act_fwd")
            cin = _f32((N, 64), dev)
            _hip.check(lib.tn_color_input_fwd(fld, d.data_ptr(), bo.data_ptr() + 4, ldb, cam.data_ptr(), 1, R, S,
                                              cin.data_ptr(), _stream()), "tn_color_input_fwd")
            c1 = linear_fwd(cin, 0, 64, fld.head0, ACT_RELU, N)
            c2 = linear_fwd(c1, 0, c1.shape[1], fld.head1, ACT_RELU, N)
            rgb_s = linear_fwd(c2, 0, c2.shape[1], fld.head2, ACT_SIGMOID, N)
            t1 = linear_fwd(bo, 1, ldb, fld.th0, ACT_RELU, N)
            t2 = linear_fwd(t1, 0, t1.shape[1], fld.th1, ACT_SIGMOID, N)
            th_s = linear_fwd(t2, 0, t2.shape[1], fld.thead, ACT_NONE, N)
        # get_weights + the RGB / thermal / accumulation renderers of the level: one launch (tn_ray_render_fwd)
        f.weights = _f32((R, S), dev)
        rgb, thermal, acc = _f32((R, 3), dev), _f32((R, 1), dev), _f32((R, 1), dev)
        _hip.check(lib.tn_ray_render_fwd(f.deltas.data_ptr(), f.density.data_ptr(), rgb_s.data_ptr(), th_s.data_ptr(), R, S,
                                         f.weights.data_ptr(), rgb.data_ptr(), thermal.data_ptr(), acc.data_ptr(), _stream()),
                   "tn_ray_render_fwd")
        # the step's regularisers start now, beside the depth renderers (config.overlap_regularisers)
        _precompute_regularisers(model, [t.weights for t in tapes] + [f.weights], [t.spacing for t in tapes] + [f.spacing])
        depth, expected = _f32((R, 1), dev), _f32((R, 1), dev)
        scratch = _f32((2,), dev)
        starts, ends = _starts_ends(f)
        _hip.check(lib.tn_depth_fwd(f.weights.data_ptr(), starts.data_ptr(), ends.data_ptr(), R, S, None,
                                    depth.data_ptr(), expected.data_ptr(), scratch.data_ptr(), _stream()), "tn_depth_fwd")
        for t in (tapes if not prop_depths else []):
            pd = _f32((R, 1), dev)
            st, en = _starts_ends(t)
            _hip.check(lib.tn_depth_fwd(t.weights.data_ptr(), st.data_ptr(), en.data_ptr(), R, t.weights.shape[1], None,
                                        pd.data_ptr(), None, None, _stream()), "tn_depth_fwd")
            prop_depths.append(pd)

        ctx.set_materialize_grads(False)  # outputs nobody differentiates arrive as None in backward, not as zero tensors
        ctx.model, ctx.tapes, ctx.field_tape = model, tapes, f
        ctx.acts = (h1, bo, cin, c1, c2, rgb_s, t1, t2, th_s)
        ctx.acc, ctx.o, ctx.d, ctx.cam = acc, o, d, cam
        ctx.updated = bool(updated)
        ctx.tape_free = tape_free
        ctx.param_names = model.named_parameter_lists()[0]
        ctx.params = dict(zip(ctx.param_names, params))
        # ctx must not hold a tensor OBJECT that is also returned as a differentiable output (output -> grad_fn -> ctx ->
        # output is a cycle the collector cannot see: the step's whole tape would leak); return fresh views instead
        outs = (rgb, thermal, acc.view(R, 1), tapes[0].weights[..., None], tapes[1].weights[..., None], f.weights[..., None],
                depth, expected, prop_depths[0], prop_depths[1], tapes[0].spacing, tapes[1].spacing, f.spacing,
                tapes[0].eucl, tapes[1].eucl, f.eucl)
        ctx.mark_non_differentiable(*outs[6:])
        if not ctx.updated:  # NS evaluates the proposal densities under no_grad on these steps
            ctx.mark_non_differentiable(outs[3], outs[4])
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_th, g_acc, g_w0, g_w1, g_w2, *unused):
        if getattr(ctx, "step_call", None) is not None:
            return _StepCall.backward(ctx, g_rgb, g_th, g_acc, g_w2)
        lib = _hip.load()
        model, f = ctx.model, ctx.field_tape
        cfg = model.config
        dev = ctx.o.device
        _drop_precomputed((dev, _hip.current_stream()))  # side-stream regularisers of this forward that no loss collected
        h1, bo, cin, c1, c2, rgb_s, t1, t2, th_s = ctx.acts
        R, S = f.weights.shape
        N = R * S
        fld = model.field.train_struct()
        like = ctx.params
        grads: Dict[str, Tensor] = {}
        # camera-pose optimisation: the ray origins / directions carry gradient (NS CameraOptimizer.apply_to_raybundle)
        ray_grads = None
        R_, S_ = f.weights.shape
        arena = _GradArena(like, dev, extra=R_ * (64 + 2 * 64 + S_) + 1024)  # every gradient buffer of this step: one allocation, one fill
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            ray_grads = (arena.zeros(tuple(ctx.o.shape)), arena.zeros(tuple(ctx.d.shape)))
        # NS SHEncoding.pytorch_fwd is @torch.no_grad(): by default the SH basis passes no gradient to the
        # directions; config.sh_direction_gradient=True adds that term (a differentiable SH encoding)
        sh_grads = ray_grads is not None and bool(cfg.sh_direction_gradient)

        def zeros(name: str) -> Tensor:
            grads[name] = arena.get(name)
            return grads[name]

        chained = bool(cfg.fused_train_backward)  # each MLP's layers in ONE launch (tn_linear_chain_bwd) vs one launch per layer
        bucketed = bool(getattr(cfg, "bucketed_table_scatter", True))
        spread = bool(getattr(cfg, "spread_coarse_scatter", True))
        exp_min = float(getattr(cfg, "trunc_exp_clamp_min", -15.0))
        # ---- proposal levels, update steps (1 in 6 after warm-up) ---------------------------------------------------
        # Their chain (get_weights adjoint -> density MLP -> the proposal grids' scatter -> positions) shares nothing with the final
        # level's but the ray gradients: queued on the step's third stream it runs beside the MFMA-bound field backward (its own
        # ray-gradient buffers, added at the join).  Only the fused forms qualify: the stage chain issues torch ops.
        prop_join = None
        g_prop = (g_w0, g_w1)
        if ctx.updated and any(g is not None for g in g_prop):
            fused_levels = all(t.hid is None for t in ctx.tapes)
            if fused_levels and getattr(cfg, "overlap_table_scatter", True):
                main, _, third = _step_streams(dev)
                ray_p = (arena.zeros(tuple(ctx.o.shape)), arena.zeros(tuple(ctx.d.shape))) if ray_grads else None
                gs = [None if g is None else g.reshape(t.weights.shape).contiguous() for g, t in zip(g_prop, ctx.tapes)]
                keep: list = []
                third.wait_stream(main)  # the cleared arena and the interlevel gradients are the main stream's work so far
                _STREAM_OVERRIDE[0] = third.cuda_stream  # launches and allocations only until it is cleared
                try:
                    RenderTrain._proposal_levels(ctx, model, grads, arena, ray_p, chained, bucketed, exp_min, gs, keep)
                finally:
                    _STREAM_OVERRIDE[0] = None
                prop_join = (third, ray_p, keep)
                g_prop = (None, None)

        # ---- final level ------------------------------------------------------------------------------------
        # adjoints of the level's renderers and of get_weights (+ use_gradient_scaling, REF :228-231): one launch
        g_rgb_s = _f32((N, 3), dev) if g_rgb is not None else None
        g_th_s = _f32((N, 1), dev) if g_th is not None else None
        g_density = _f32((R, S), dev)
        g_wx = None if g_w2 is None else _hip.require_device_tensor(g_w2.reshape(R, S), "d weights")
        st_en = _starts_ends(f) if cfg.use_gradient_scaling else (None, None)
        _hip.check(lib.tn_ray_render_bwd(f.deltas.data_ptr(), f.density.data_ptr(), rgb_s.data_ptr(), th_s.data_ptr(),
                                         ctx.acc.data_ptr(), _hip.ptr(None if g_rgb is None else g_rgb.contiguous()),
                                         _hip.ptr(None if g_th is None else g_th.contiguous()),
                                         _hip.ptr(None if g_acc is None else g_acc.contiguous()), _hip.ptr(g_wx),
                                         _hip.ptr(st_en[0]), _hip.ptr(st_en[1]), R, S, _hip.ptr(g_rgb_s), _hip.ptr(g_th_s),
                                         g_density.data_ptr(), _stream()), "tn_ray_render_bwd")
        E = f.enc.shape[1]
        g_enc = _f32((N, E), dev)  # row-major [N,32] in both forms (what the table scatter reads)
        if ctx.tape_free:
            self_bias = bo  # ray_bias [R,64]
            gr = _hip.tn_field_grads()
            names = {"base0": "field.mlp_base.mlp.layers.0", "base1": "field.mlp_base.mlp.layers.1",
                     "head0": "field.mlp_head.layers.0", "head1": "field.mlp_head.layers.1", "head2": "field.mlp_head.layers.2",
                     "th0": "field.mlp_thermal.layers.0", "th1": "field.mlp_thermal.layers.1", "thead": "field.field_head_thermal.net"}
            for key, name in names.items():
                if key.startswith("head") and g_rgb_s is None:
                    continue
                if key.startswith("th") and g_th_s is None:
                    continue
                setattr(gr, key + "_w", zeros(name + ".weight").data_ptr())
                if key != "head0":
                    setattr(gr, key + "_b", zeros(name + ".bias").data_ptr())
            g_ray = arena.zeros((R, 64)) if g_rgb_s is not None else None
            ws = _fused_bwd_workspace(dev, R, S)
            # 0: one launch; 1: colour head | thermal head | mlp_base; 2: the same with the heads' 64 x 64 products as bf16 pieces
            split_form = 0 if not getattr(cfg, "fused_backward_split", True) else (2 if getattr(cfg, "backward_bf16_pieces", True) else 1)
            g_pos = _f32((N, 3), dev) if ray_grads else None  # d loss / d sample position, written by the mlp_base launch
            _hip.check(lib.tn_field_bwd_fused(fld, R, S, f.enc.data_ptr(), f.sel.data_ptr(), _hip.ptr(h1), self_bias.data_ptr(), rgb_s.data_ptr(),
                                              _hip.ptr(g_rgb_s), _hip.ptr(g_th_s), g_density.data_ptr(),
                                              1 if model.field.pass_thermal_gradients else 0, exp_min,
                                              split_form, g_enc.data_ptr(),
                                              _hip.ptr(g_ray), f.pos.data_ptr() if ray_grads else None, _hip.ptr(cin), _hip.ptr(g_pos),
                                              C.byref(gr), ws.data_ptr(), ws.numel(), _stream()),
                       "tn_field_bwd_fused")
            zeros("field.mlp_base.encoder.hash_table")
            if g_ray is not None:
                for name in ("field.mlp_head.layers.0.bias", "field.embedding_appearance.embedding.weight"):
                    zeros(name)
            g_cin = _f32((R, 64), dev) if g_ray is not None and sh_grads else None

            def ray_level_adjoints() -> None:  # nothing here touches the table gradient: queued beside its scatter
                if g_ray is not None:
                    # mlp_head.0's ray-level part: bias, SH and appearance weight columns, the embedding gradient
                    _hip.check(lib.tn_ray_head_bwd(fld, ctx.d.data_ptr(), ctx.cam.data_ptr(), R, g_ray.data_ptr(),
                                                   grads["field.mlp_head.layers.0.weight"].data_ptr(),
                                                   grads["field.mlp_head.layers.0.bias"].data_ptr(),
                                                   grads["field.embedding_appearance.embedding.weight"].data_ptr(), _hip.ptr(g_cin),
                                                   _stream()), "tn_ray_head_bwd")
                    if sh_grads:  # ... and on through the SH basis to the directions (camera-pose optimisation, differentiable-SH switch)
                        _hip.check(lib.tn_color_input_bwd(fld, g_cin.data_ptr(), ctx.cam.data_ptr(), 1, R, 1, None, 0, None,
                                                          ctx.d.data_ptr(), ray_grads[1].data_ptr(), _stream()), "tn_color_input_bwd")
                if ray_grads:
                    starts, ends = _starts_ends(f)
                    _hip.check(lib.tn_frustum_positions_bwd(g_pos.data_ptr(), starts.data_ptr(), ends.data_ptr(), R, S,
                                                            ray_grads[0].data_ptr(), ray_grads[1].data_ptr(), _stream()),
                               "tn_frustum_positions_bwd")

            # config.deferred_table_update: both halves on side streams, not joined here (steps whose third stream carries the
            # proposal levels' backward keep the joined form: that chain and the atomic half would queue behind one another)
            defer = bool(getattr(cfg, "deferred_table_update", False)) and prop_join is None
            if hash_encode_bwd(fld.grid, fld.space, f.pos, g_enc, grads["field.mlp_base.encoder.hash_table"], bucketed, spread,
                               getattr(cfg, "overlap_table_scatter", True), ray_level_adjoints, defer=defer):
                _hip.defer(dev, (), [arena.flat, f.enc, g_density, g_rgb_s, g_th_s])
            return RenderTrain._finish(ctx, model, grads, arena, ray_grads, chained, bucketed, exp_min, g_prop, prop_join)
        ldb = bo.shape[1]
        # layer widths as built (config.hidden_dim / hidden_dim_color / hidden_dim_transient; 64 in the reference's configs)
        Wb, Wc = h1.shape[1], c1.shape[1]
        Wt1, Wt2 = t1.shape[1], t2.shape[1]
        if model.field.staged:
            chained = False  # (the chained launch tiles 64-wide layers)
        g_bo = _f32((N, ldb), dev)  # column 0 written, the geo columns cleared (the += target of both heads) in one pass
        _hip.check(lib.tn_density_act_bwd(bo.data_ptr(), ldb, f.sel.data_ptr(), fld.average_init_density, exp_min,
                                          g_density.data_ptr(), N, g_bo.data_ptr(), ldb, ldb, _stream()), "tn_density_act_bwd")
        if g_th_s is not None:  # thermal branch [REF thermal_field.py:170-179]
            into_geo = g_bo if model.field.pass_thermal_gradients else None  # REF :171-172 (.detach())
            th = [(fld.thead, t2, 0, Wt2, ACT_SIGMOID, zeros("field.field_head_thermal.net.weight"),
                   zeros("field.field_head_thermal.net.bias")),
                  (fld.th1, t1, 0, Wt1, ACT_RELU, zeros("field.mlp_thermal.layers.1.weight"), zeros("field.mlp_thermal.layers.1.bias")),
                  (fld.th0, bo, 1, ldb, ACT_NONE, zeros("field.mlp_thermal.layers.0.weight"), zeros("field.mlp_thermal.layers.0.bias"))]
            if chained:
                linear_chain_bwd(th, None, ACT_NONE, g_th_s, 1, N, into_geo, 1, ldb, True)
            else:
                g_t2, g_t1 = _f32((N, Wt2), dev), _f32((N, Wt1), dev)
                linear_bwd(t2, 0, Wt2, None, g_th_s, 1, fld.thead, ACT_NONE, N, g_t2, 0, Wt2, False, th[0][5], th[0][6])
                linear_bwd(t1, 0, Wt1, t2, g_t2, Wt2, fld.th1, ACT_SIGMOID, N, g_t1, 0, Wt1, False, th[1][5], th[1][6])
                linear_bwd(bo, 1, ldb, t1, g_t1, Wt1, fld.th0, ACT_RELU, N, into_geo, 1, ldb, True, th[2][5], th[2][6])
        if g_rgb_s is not None:  # colour branch [REF :160-168]
            g_cin = _f32((N, 64), dev)
            hd = [(fld.head2, c2, 0, Wc, ACT_RELU, zeros("field.mlp_head.layers.2.weight"), zeros("field.mlp_head.layers.2.bias")),
                  (fld.head1, c1, 0, Wc, ACT_RELU, zeros("field.mlp_head.layers.1.weight"), zeros("field.mlp_head.layers.1.bias")),
                  (fld.head0, cin, 0, 64, ACT_NONE, zeros("field.mlp_head.layers.0.weight"), zeros("field.mlp_head.layers.0.bias"))]
            if chained:
                linear_chain_bwd(hd, rgb_s, ACT_SIGMOID, g_rgb_s, 3, N, g_cin, 0, 64, False)
            else:
                g_c2, g_c1 = _f32((N, Wc), dev), _f32((N, Wc), dev)
                linear_bwd(c2, 0, Wc, rgb_s, g_rgb_s, 3, fld.head2, ACT_SIGMOID, N, g_c2, 0, Wc, False, hd[0][5], hd[0][6])
                linear_bwd(c1, 0, Wc, c2, g_c2, Wc, fld.head1, ACT_RELU, N, g_c1, 0, Wc, False, hd[1][5], hd[1][6])
                linear_bwd(cin, 0, 64, c1, g_c1, Wc, fld.head0, ACT_RELU, N, g_cin, 0, 64, False, hd[2][5], hd[2][6])
            _hip.check(lib.tn_color_input_bwd(fld, g_cin.data_ptr(), ctx.cam.data_ptr(), 1, R, S, g_bo.data_ptr() + 4, ldb,
                                              zeros("field.embedding_appearance.embedding.weight").data_ptr(),
                                              ctx.d.data_ptr() if sh_grads else None,
                                              ray_grads[1].data_ptr() if sh_grads else None, _stream()),
                       "tn_color_input_bwd")
        bs = [(fld.base1, h1, 0, Wb, ACT_RELU, zeros("field.mlp_base.mlp.layers.1.weight"), zeros("field.mlp_base.mlp.layers.1.bias")),
              (fld.base0, f.enc, 0, E, ACT_NONE, zeros("field.mlp_base.mlp.layers.0.weight"), zeros("field.mlp_base.mlp.layers.0.bias"))]
        if chained:
            linear_chain_bwd(bs, None, ACT_NONE, g_bo, ldb, N, g_enc, 0, E, False)
        else:
            g_h1 = _f32((N, Wb), dev)
            linear_bwd(h1, 0, Wb, None, g_bo, ldb, fld.base1, ACT_NONE, N, g_h1, 0, Wb, False, bs[0][5], bs[0][6])
            linear_bwd(f.enc, 0, E, h1, g_h1, Wb, fld.base0, ACT_RELU, N, g_enc, 0, E, False, bs[1][5], bs[1][6])
        hash_encode_bwd(fld.grid, fld.space, f.pos, g_enc, zeros("field.mlp_base.encoder.hash_table"), bucketed, spread,
                        getattr(cfg, "overlap_table_scatter", True))
        if ray_grads:
            _ray_grads_from_enc(fld.grid, fld.space, f, g_enc, *ray_grads)
        return RenderTrain._finish(ctx, model, grads, arena, ray_grads, chained, bucketed, exp_min, g_prop, prop_join)

    @staticmethod
    def _proposal_levels(ctx, model, grads, arena, ray_grads, chained, bucketed, exp_min, g_prop, keep=None) -> None:
        """backward of the proposal levels (they receive gradient only through their weights: the interlevel loss)"""
        for lvl, g in enumerate(g_prop):
            if g is None:
                continue
            t = ctx.tapes[lvl]
            which = min(lvl, len(model.proposal_networks) - 1)  # one shared network: both levels accumulate into it
            net = model.proposal_networks[which].train_struct()
            _proposal_level_bwd(net, t, g.reshape(t.weights.shape).contiguous(), grads, f"proposal_networks.{which}", arena,
                                ray_grads, chained, bucketed, exp_min, bool(getattr(model.config, "spread_coarse_scatter", True)), keep)

    @staticmethod
    def _finish(ctx, model, grads, arena, ray_grads, chained, bucketed, exp_min, g_prop, prop_join=None):
        """proposal levels unless they are already queued on the third stream (then: the join), then the gradient tuple in
        parameter order"""
        if ctx.updated:
            RenderTrain._proposal_levels(ctx, model, grads, arena, ray_grads, chained, bucketed, exp_min, g_prop)
        if prop_join is not None:
            third, ray_p, keep = prop_join
            _step_streams(ctx.o.device)[0].wait_stream(third)
            if ray_grads and ray_p:
                ray_grads[0].add_(ray_p[0])
                ray_grads[1].add_(ray_p[1])
            keep.clear()

        g_o, g_d = ray_grads if ray_grads else (None, None)
        result = (None, g_o, g_d) + (None,) * 5 + tuple(grads.get(n) for n in ctx.param_names)
        ctx.tapes = ctx.field_tape = ctx.acts = ctx.acc = None  # the tape is dead after one backward
        return result


# --------------------------------------------------------------------------------------------------
# the step's launch chains as two C-ABI calls (tn_train_step_fwd / tn_train_step_bwd)
# --------------------------------------------------------------------------------------------------
def _step_call_applies(model, cfg) -> bool:
    """config.fused_step_calls (default on) on the default training configuration: tape-free final level, split backward; any
    other setting keeps the per-call path above (which is also the cross-check: same launches, same streams, same bits)."""
    return bool(getattr(cfg, "fused_step_calls", True)) and bool(getattr(cfg, "tape_free_training", True)) \
        and bool(getattr(cfg, "fused_backward_split", True)) and cfg.num_proposal_iterations == 2 and not model.field.staged


def _al(n: int) -> int:
    return (n + 63) // 64 * 64  # 256-byte aligned carve-outs


class _StepCall:
    """RenderTrain's forward / backward on a step whose proposal networks take no gradient, each as ONE library call.  The
    per-sample tensors of the step live in one slab (they are only ever passed on as addresses), the per-ray outputs in another
    (the tensors autograd hands back to the caller must not keep ~150 B per sample alive)."""

    @staticmethod
    def forward(ctx, model, o, d, nears, fars, cam, jitter, params, prop_structs, anneal, uniform, single):
        lib = _hip.load()
        cfg = model.config
        dev = o.device
        R = o.shape[0]
        P0, P1 = cfg.num_proposal_samples_per_ray
        S = cfg.num_nerf_samples_per_ray
        N, NP = R * S, (R * S + 63) // 64 * 64
        hit = model.field.train_struct(prepare="struct")
        if hit is None:
            raise RuntimeError("config.fused_step_calls needs the reference field geometry (the MFMA chain); set it to False")
        raw, prepared, prepared_bytes = hit
        main, second, third = _step_streams(dev)
        rays_grad = o.requires_grad or d.requires_grad
        keep_base = bool(getattr(cfg, "store_base_output", True))
        keep_jac = rays_grad and bool(getattr(cfg, "store_position_jacobian", True))
        # ---- per-ray slab: everything that leaves as a tensor --------------------------------------------------------------------
        ns = (P0, P1, S)
        shapes = [("rgb", (R, 3)), ("thermal", (R, 1)), ("acc", (R, 1)), ("depth", (R, 1)), ("expected", (R, 1)), ("pd0", (R, 1)),
                  ("pd1", (R, 1))]
        for i, n in enumerate(ns):
            shapes += [("w%d" % i, (R, n)), ("sp%d" % i, (R, n + 1)), ("eu%d" % i, (R, n + 1))]
        want = getattr(cfg, "overlap_regularisers", "auto")
        if want == "auto":
            want = N >= 4096 * 96
        mult_d, mult_i = float(cfg.distortion_loss_mult), float(cfg.interlevel_loss_mult)
        if want:
            shapes += [("g_dist", (R, S)), ("g_i0", (R, P0)), ("g_i1", (R, P1))]
        off, total = {}, 0
        for name, shp in shapes:
            off[name] = total
            total += _al(shp[0] * shp[1])
        ray_slab = torch.empty((total,), dtype=torch.float32, device=dev)
        t = {name: ray_slab[off[name]:off[name] + shp[0] * shp[1]].view(shp) for name, shp in shapes}
        # ---- per-sample slab: addresses only -------------------------------------------------------------------------------------
        need_ws = 0
        rc = _hip.tn_render_config()
        rc.num_proposal_samples[0], rc.num_proposal_samples[1], rc.num_nerf_samples = P0, P1, S
        rc.training, rc.pdf_anneal, rc.early_stop_transmittance, rc.kernel_family = 1, anneal, 0.0, 0
        rc.initial_sampler = uniform
        rc.per_sample_jitter = 0 if single else 1
        need_ws = lib.tn_render_workspace_bytes(rc, R)
        sizes = [("pos", N * 3), ("starts", N), ("ends", N), ("deltas", N), ("ray_bias", R * 64), ("enc", NP * 32), ("sel", N),
                 ("density", N), ("rgb_s", N * 3), ("th_s", N), ("base_out", N * 16 if keep_base else 0),
                 ("jac", NP * 96 if keep_jac else 0), ("scratch", 2 * ((R + 3) // 4)), ("ws", (need_ws + 3) // 4)]
        so, stotal = {}, 0
        for name, n in sizes:
            so[name] = stotal
            stotal += _al(n)
        slab = torch.empty((stotal,), dtype=torch.float32, device=dev)
        base = slab.data_ptr()

        def ptr(name):
            return base + 4 * so[name]

        st = _hip.tn_train_step()
        st.prop0, st.prop1 = C.pointer(prop_structs[0]), C.pointer(prop_structs[1])
        st.field_raw, st.field, st.prepared_bytes = C.pointer(raw), C.pointer(prepared), prepared_bytes
        ins = _hip.tn_render_inputs()
        ins.origins, ins.directions, ins.nears, ins.fars = o.data_ptr(), d.data_ptr(), nears.data_ptr(), fars.data_ptr()
        ins.camera_indices, ins.jitter = cam.data_ptr(), jitter.data_ptr()
        ins.lin_bins0 = linspace_bins(P0, dev).data_ptr()
        ins.u1 = pdf_positions(P1 + 1, dev, True).data_ptr()
        ins.u2 = pdf_positions(S + 1, dev, True).data_ptr()
        st.cfg, st.inputs, st.num_rays = C.pointer(rc), C.pointer(ins), R
        for i in range(3):
            st.spacing[i], st.eucl[i], st.weights[i] = t["sp%d" % i].data_ptr(), t["eu%d" % i].data_ptr(), t["w%d" % i].data_ptr()
        st.prop_depth[0], st.prop_depth[1] = t["pd0"].data_ptr(), t["pd1"].data_ptr()
        st.positions, st.starts, st.ends, st.deltas, st.ray_bias = ptr("pos"), ptr("starts"), ptr("ends"), ptr("deltas"), ptr("ray_bias")
        st.enc, st.selector, st.density, st.rgb_samples, st.thermal_samples = ptr("enc"), ptr("sel"), ptr("density"), ptr("rgb_s"), ptr("th_s")
        st.base_out = ptr("base_out") if keep_base else None
        st.jacobian = ptr("jac") if keep_jac else None
        st.rgb, st.thermal, st.accumulation = t["rgb"].data_ptr(), t["thermal"].data_ptr(), t["acc"].data_ptr()
        st.depth, st.expected_depth, st.depth_scratch = t["depth"].data_ptr(), t["expected"].data_ptr(), ptr("scratch")
        st.workspace, st.workspace_bytes = ptr("ws"), need_ws
        # the backward's gradient arena: allocated now, cleared by the call on the second stream
        arena = _GradArena(dict(zip(model.named_parameter_lists()[0], params)), dev, extra=R * (64 + 2 * 64 + S) + 1024, zero=False)
        st.zero_buffer, st.zero_bytes = arena.flat.data_ptr(), arena.flat.numel() * 4
        # (the backward joins the second stream before it touches the arena; a training-mode forward that is never differentiated
        # drops the arena with the fill possibly still queued: the allocator must know the second stream used the block)
        arena.flat.record_stream(second)
        slot = (dev, _hip.current_stream())
        _drop_precomputed(slot)
        entry = None
        if want:
            st.distortion_mult, st.interlevel_mult = mult_d, mult_i
            entry = {"hold": (ray_slab,)}
            w2, c2 = t["w2"], t["sp2"]
            if mult_d:
                loss_d = _hip.fresh_zeros((2,), dev)
                st.distortion_loss_pair, st.distortion_grad = loss_d.data_ptr(), t["g_dist"].data_ptr()
                entry["dist"] = ((_tensor_key(w2), _tensor_key(c2), mult_d), (loss_d, t["g_dist"]), second)
            loss_i = _hip.fresh_zeros((1,), dev)
            st.interlevel_loss = loss_i.data_ptr()
            st.interlevel_grad[0], st.interlevel_grad[1] = t["g_i0"].data_ptr(), t["g_i1"].data_ptr()
            key = (_tensor_key(w2), _tensor_key(c2), mult_i) + tuple((_tensor_key(t["w%d" % i]), _tensor_key(t["sp%d" % i])) for i in range(2))
            entry["inter"] = (key, (loss_i, [t["g_i0"], t["g_i1"]]), third)
        st.stream, st.second, st.third = main.cuda_stream, second.cuda_stream, third.cuda_stream
        # a deferred table update of the previous step (config.deferred_table_update): the call waits for it right before the
        # field launch; the registry entry (and the temporaries it holds) is released here — the events carry the order
        pend = _hip.take_pending(dev)
        events = []
        if pend is not None:
            for s_ in pend["streams"]:
                if s_.cuda_stream != main.cuda_stream:
                    ev = torch.cuda.Event()
                    ev.record(s_)
                    events.append(ev)
        if events:
            arr = (C.c_void_p * len(events))(*[ev.cuda_event for ev in events])
            st.wait_events, st.num_wait_events = arr, len(events)
        _hip.check(lib.tn_train_step_fwd(C.byref(st)), "tn_train_step_fwd")
        # (the deferred update's temporaries — pend["keep"] — may go now: whatever reuses their memory is queued on the calling
        # stream behind the events just waited for)
        pend = None
        if entry is not None:
            _REG_PRE[slot] = entry

        ctx.set_materialize_grads(False)
        ctx.model, ctx.o, ctx.d, ctx.cam = model, o, d, cam
        ctx.step_call = (slab, so, ray_slab, off, R, S, keep_base, keep_jac, arena)
        ctx.updated, ctx.tape_free = False, True
        ctx.tapes = ctx.field_tape = ctx.acts = ctx.acc = None
        ctx.param_names = model.named_parameter_lists()[0]
        ctx.params = dict(zip(ctx.param_names, params))
        # (fresh views: ctx must not hold a tensor OBJECT that is also returned, see RenderTrain.forward)
        outs = (t["rgb"], t["thermal"], t["acc"], t["w0"][..., None], t["w1"][..., None], t["w2"][..., None], t["depth"], t["expected"],
                t["pd0"], t["pd1"], t["sp0"], t["sp1"], t["sp2"], t["eu0"], t["eu1"], t["eu2"])
        ctx.mark_non_differentiable(*outs[3:5], *outs[6:])
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_th, g_acc, g_w2):
        lib = _hip.load()
        model = ctx.model
        cfg = model.config
        slab, so, ray_slab, off, R, S, keep_base, keep_jac, arena = ctx.step_call
        dev = slab.device
        N = R * S
        _drop_precomputed((dev, _hip.current_stream()))
        fld = model.field.train_struct()
        ray_grads = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            ray_grads = (arena.zeros(tuple(ctx.o.shape)), arena.zeros(tuple(ctx.d.shape)))
        sh_grads = ray_grads is not None and bool(cfg.sh_direction_gradient)
        grads: Dict[str, Tensor] = {}
        base, rbase = slab.data_ptr(), ray_slab.data_ptr()

        def ptr(name):
            return base + 4 * so[name]

        a = _hip.tn_train_step_bwd_args()
        a.field, a.num_rays, a.n = C.pointer(fld), R, S
        a.positions, a.starts, a.ends, a.deltas, a.ray_bias = ptr("pos"), ptr("starts"), ptr("ends"), ptr("deltas"), ptr("ray_bias")
        a.enc, a.selector, a.density, a.rgb_samples, a.thermal_samples = ptr("enc"), ptr("sel"), ptr("density"), ptr("rgb_s"), ptr("th_s")
        a.base_out = ptr("base_out") if keep_base else None
        a.jacobian = ptr("jac") if keep_jac else None
        a.accumulation = rbase + 4 * off["acc"]
        a.directions, a.camera_indices = ctx.d.data_ptr(), ctx.cam.data_ptr()
        hold = []  # contiguous copies of the incoming gradients live until the call has been queued (same stream: no race)
        for name, g in (("d_rgb", g_rgb), ("d_thermal", g_th), ("d_accumulation", g_acc), ("d_weights", g_w2)):
            if g is not None:
                g = _hip.require_device_tensor(g, name)
                hold.append(g)
                setattr(a, name, g.data_ptr())
        a.use_gradient_scaling = 1 if cfg.use_gradient_scaling else 0
        a.pass_thermal_gradients = 1 if model.field.pass_thermal_gradients else 0
        a.split_form = 2 if getattr(cfg, "backward_bf16_pieces", True) else 1
        a.sh_direction_gradient = 1 if sh_grads else 0
        a.trunc_exp_min = float(getattr(cfg, "trunc_exp_clamp_min", -15.0))
        # scratch of the backward: one slab (addresses only)
        first = -1
        bucketed = getattr(cfg, "bucketed_table_scatter", True)
        if bucketed is True:
            first = lib.tn_hash_encode_bwd_sorted_first_level(fld.grid, N)
        elif bucketed is not False and bucketed is not None:
            first = int(bucketed)  # (tests, tools/train_bench.py --first-sorted-level: bucketed from that level on)
        need_sorted = lib.tn_hash_encode_bwd_sorted_workspace_bytes(fld.grid, N, first) if first >= 0 else 0
        sizes = [("g_rgb_s", N * 3 if g_rgb is not None else 0), ("g_th_s", N if g_th is not None else 0), ("g_density", N), ("g_enc", N * 32),
                 ("g_pos", N * 3 if ray_grads else 0), ("g_cin", R * 64 if (sh_grads and g_rgb is not None) else 0),
                 ("sorted", (need_sorted + 3) // 4)]
        bo, btotal = {}, 0
        for name, n in sizes:
            bo[name] = btotal
            btotal += _al(n)
        try:
            bslab = torch.empty((btotal,), dtype=torch.float32, device=dev)
        except torch.cuda.OutOfMemoryError:  # no room for the records: the same sums through the global atomics
            btotal -= _al(sizes[-1][1])
            bslab, first, need_sorted = torch.empty((btotal,), dtype=torch.float32, device=dev), -1, 0
        bb = bslab.data_ptr()
        a.d_rgb_samples = bb + 4 * bo["g_rgb_s"] if g_rgb is not None else None
        a.d_thermal_samples = bb + 4 * bo["g_th_s"] if g_th is not None else None
        a.d_density, a.d_enc = bb + 4 * bo["g_density"], bb + 4 * bo["g_enc"]
        a.d_positions = bb + 4 * bo["g_pos"] if ray_grads else None
        a.d_ray_inputs = bb + 4 * bo["g_cin"] if (sh_grads and g_rgb is not None) else None
        gr = _hip.tn_field_grads()
        names = {"base0": "field.mlp_base.mlp.layers.0", "base1": "field.mlp_base.mlp.layers.1",
                 "head0": "field.mlp_head.layers.0", "head1": "field.mlp_head.layers.1", "head2": "field.mlp_head.layers.2",
                 "th0": "field.mlp_thermal.layers.0", "th1": "field.mlp_thermal.layers.1", "thead": "field.field_head_thermal.net"}

        def zeros(name: str) -> Tensor:
            grads[name] = arena.get(name)
            return grads[name]

        for key, name in names.items():
            if key.startswith("head") and g_rgb is None:
                continue
            if key.startswith("th") and g_th is None:
                continue
            setattr(gr, key + "_w", zeros(name + ".weight").data_ptr())
            if key != "head0":
                setattr(gr, key + "_b", zeros(name + ".bias").data_ptr())
        a.grads = C.pointer(gr)
        if g_rgb is not None:
            a.d_ray_sum = arena.zeros((R, 64)).data_ptr()
            a.d_head0_bias = zeros("field.mlp_head.layers.0.bias").data_ptr()
            a.d_appearance = zeros("field.embedding_appearance.embedding.weight").data_ptr()
        a.d_table = zeros("field.mlp_base.encoder.hash_table").data_ptr()
        if ray_grads:
            a.d_origins, a.d_directions = ray_grads[0].data_ptr(), ray_grads[1].data_ptr()
        ws = _fused_bwd_workspace(dev, R, S)
        a.fused_workspace, a.fused_workspace_bytes = ws.data_ptr(), ws.numel()
        a.first_sorted_level = first
        if need_sorted:
            a.sorted_workspace, a.sorted_workspace_bytes = bb + 4 * bo["sorted"], need_sorted
        main, second, third = _step_streams(dev)
        overlap = bool(getattr(cfg, "overlap_table_scatter", True))
        defer = bool(getattr(cfg, "deferred_table_update", False)) and overlap and first > 0
        a.spread = 1 if getattr(cfg, "spread_coarse_scatter", True) else 0
        if a.spread:
            need = lib.tn_hash_encode_bwd_spread_workspace_bytes(fld.grid)
            if need:
                key = (dev, third.cuda_stream if defer else main.cuda_stream, need)
                sws = _SPREAD_WS.get(key)
                if sws is None:
                    sws = _SPREAD_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
                a.spread_workspace, a.spread_workspace_bytes = sws.data_ptr(), need
        a.overlap, a.defer = (1 if overlap else 0), (1 if defer else 0)
        a.wait_second_first = 1  # the arena was cleared on the second stream (tn_train_step_fwd)
        a.stream, a.second, a.third = main.cuda_stream, second.cuda_stream, third.cuda_stream
        _hip.check(lib.tn_train_step_bwd(C.byref(a)), "tn_train_step_bwd")
        if defer:  # both halves of the scatter are still out: whoever reads the table (or its gradient) joins (_hip.join_pending)
            _hip.defer(dev, [second, third], [bslab, slab, ray_slab, arena.flat])
        g_o, g_d = ray_grads if ray_grads else (None, None)
        result = (None, g_o, g_d) + (None,) * 5 + tuple(grads.get(n) for n in ctx.param_names)
        ctx.step_call = None
        return result


# --------------------------------------------------------------------------------------------------
# regularisers
# --------------------------------------------------------------------------------------------------
def _distortion_run(w: Tensor, b: Tensor, mult: float, side=None) -> Tuple[Tensor, Tensor]:
    """tn_distortion_loss_term on [R,n] weights / [R,n+1] bins -> (loss pair, gradient of the term); ``side``: a stream to run it on
    beside the calling one (its outputs are allocated and cleared on the calling stream first)"""
    R, n = w.shape
    loss = _hip.fresh_zeros((2,), w.device)
    g = _f32((R, n), w.device)
    stream = _stream()
    if side is not None:
        side.wait_stream(_step_streams(w.device)[0])
        stream = side.cuda_stream
    # the mean over rays and the loss multiplier are applied inside the kernel (loss and gradient): no elementwise launches around it
    _hip.check(_hip.load().tn_distortion_loss_term(b.data_ptr(), w.data_ptr(), R, n, 1.0 / R, mult, loss.data_ptr(), g.data_ptr(),
                                                   stream), "tn_distortion_loss_term")
    return loss, g


def _interlevel_run(mult: float, w2: Tensor, c2: Tensor, levels: Sequence[Tuple[Tensor, Tensor]], side=None) -> Tuple[Tensor, List[Tensor]]:
    """tn_interlevel_loss_levels: final level (w2 [R,n], c2 [R,n+1]) against the proposal levels [(wp [R,p], cp [R,p+1]), ...] ->
    (loss [1], one gradient per level); both levels in one launch, each scaled by mult / (R n)"""
    R, n = w2.shape
    loss = _hip.fresh_zeros((1,), w2.device)
    L = len(levels)
    cps, wps, gs, ps = (C.c_void_p * L)(), (C.c_void_p * L)(), (C.c_void_p * L)(), (C.c_int32 * L)()
    grads = []
    for k, (wp2, cp2) in enumerate(levels):
        g = _f32((R, wp2.shape[1]), wp2.device)
        cps[k], wps[k], gs[k], ps[k] = cp2.data_ptr(), wp2.data_ptr(), g.data_ptr(), wp2.shape[1]
        grads.append(g)
    stream = _stream()
    if side is not None:
        side.wait_stream(_step_streams(w2.device)[0])
        stream = side.cuda_stream
    _hip.check(_hip.load().tn_interlevel_loss_levels(c2.data_ptr(), w2.data_ptr(), R, n, L, cps, wps, ps, mult / (R * n), loss.data_ptr(),
                                                     gs, stream), "tn_interlevel_loss_levels")
    return loss, grads


# The two regularisers of a training step depend on the forward's weights and bins only, and each is a short latency-bound
# launch (22 + 31 us at S=192 for 4096 rays) that get_metrics_dict / get_loss_dict would queue one after the other behind the
# depth renderers.  With config.overlap_regularisers the training forward launches them right behind the final level's
# weights, on the step's side streams, beside the depth renderers and the image losses; _Distortion / _Interlevel pick the
# results up (same tensors, same multipliers: checked) after joining that stream.  One entry per (device, stream): the current step's.
_REG_PRE: Dict = {}


def _tensor_key(t: Tensor) -> Tuple[int, int, Tuple[int, ...]]:
    """address, in-place version counter (shared by a tensor and its views) and shape: an in-place edit between the forward and
    the loss changes the key, and the side-stream result is not served for it"""
    return (t.data_ptr(), t._version, tuple(t.shape))


def _drop_precomputed(slot) -> None:
    """forget a slot's side-stream results; the calling stream first joins the streams that write them (their outputs are the
    main stream's allocations: released unjoined, the allocator could hand them out while a side kernel is still pending)"""
    entry = _REG_PRE.pop(slot, None)
    if entry is None:
        return
    main = _step_streams(slot[0])[0]
    for which in ("dist", "inter"):
        if which in entry:
            main.wait_stream(entry[which][2])


def _precompute_regularisers(model, w_levels: Sequence[Tensor], c_levels: Sequence[Tensor]) -> None:
    cfg = model.config
    dev = w_levels[-1].device
    slot = (dev, _hip.current_stream())
    _drop_precomputed(slot)  # an earlier forward's results nobody collected (an exception, a forward without losses)
    want = getattr(cfg, "overlap_regularisers", "auto")
    if want == "auto":
        # the side launches cost the host four stream joins (~40 us): they pay once the step's device time is well above its
        # host time — S=192: 2.64 against 2.68 ms per step; S=48 (device 1.34 ms, host ~1 ms): 1.34-1.46 against 1.34-1.35
        want = w_levels[-1].numel() >= 4096 * 96
    if not want or len(w_levels) < 2:
        return
    _, second, third = _step_streams(dev)
    w2, c2 = w_levels[-1], c_levels[-1]
    entry = {"hold": (list(w_levels), list(c_levels))}  # the inputs stay alive (and their addresses theirs) while the entry exists
    mult_d = float(cfg.distortion_loss_mult)
    if mult_d:
        entry["dist"] = ((_tensor_key(w2), _tensor_key(c2), mult_d), _distortion_run(w2, c2, mult_d, second), second)
    mult_i = float(cfg.interlevel_loss_mult)
    levels = list(zip(w_levels[:-1], c_levels[:-1]))
    key = (_tensor_key(w2), _tensor_key(c2), mult_i) + tuple((_tensor_key(wp), _tensor_key(cp)) for wp, cp in levels)
    entry["inter"] = (key, _interlevel_run(mult_i, w2, c2, levels, third), third)
    _REG_PRE[slot] = entry


def _take_precomputed(dev, which: str, key):
    """the side-stream result of this step's forward for exactly these inputs (same storage, shape, in-place version and
    multiplier), joined into the calling stream — or None, and then the entry stays for a caller it does match"""
    slot = (dev, _hip.current_stream())
    entry = _REG_PRE.get(slot)
    if entry is None or which not in entry or entry[which][0] != key:
        return None
    _, result, side = entry.pop(which)
    _step_streams(dev)[0].wait_stream(side)
    if not any(x in entry for x in ("dist", "inter")):
        _REG_PRE.pop(slot, None)
    return result


# ---- the step's total loss and its backward seed --------------------------------------------------------------------------------
# nerfstudio's Trainer sums the loss dictionary with torch.add and calls backward() on the sum; as torch ops that is len - 1 add
# launches, a ones_like fill for the seed, and — in the loss Functions below, whose gradients are made by the forward launch — one
# `seed * gradient` launch per loss term: eleven 4-6 us launches per step between the forward and the backward.
# total_loss() sums the terms in ONE launch; backward_total() seeds the backward with a cached ones tensor that the loss Functions
# RECOGNISE (same storage): a unit seed needs no multiplication, so their stored gradients pass through untouched.  A plain
# total.backward() — or any other seed — takes the general path and gives the same numbers.
_UNIT_SEEDS: Dict = {}


def _unit_seed(dev) -> Tensor:
    t = _UNIT_SEEDS.get(dev)
    if t is None:
        t = _UNIT_SEEDS[dev] = torch.ones((), dtype=torch.float32, device=dev)
    return t


def _is_unit_seed(g: Optional[Tensor]) -> bool:
    if g is None or g.dim() != 0:
        return False
    t = _UNIT_SEEDS.get(g.device)
    return t is not None and g.data_ptr() == t.data_ptr()


def _seeded(go: Tensor, grad: Tensor) -> Tensor:
    """go * grad; the stored gradient itself under the unit seed of backward_total()"""
    return grad if _is_unit_seed(go) else go * grad


class _TotalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *terms: Tensor):
        ctx.n = len(terms)
        if len(terms) <= 16 and all(t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 for t in terms):
            # ((t0 + t1) + t2) + ... as torch.add would form it, in one launch (tn_sum_scalars) instead of a stack and a reduction
            out = _f32((), terms[0].device)
            ptrs = (C.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
            _hip.check(_hip.load().tn_sum_scalars(ptrs, len(terms), out.data_ptr(), _stream()), "tn_sum_scalars")
            return out
        return torch.stack([t.reshape(()) for t in terms]).sum()

    @staticmethod
    def backward(ctx, go):
        return (go,) * ctx.n


def total_loss(loss_dict: Dict[str, Tensor]) -> Tensor:
    """The sum of a loss dictionary [NS Trainer.train_iteration: functools.reduce(torch.add, loss_dict.values())] as one
    autograd node (two launches instead of one per term)."""
    terms = list(loss_dict.values())
    if len(terms) == 1:
        return terms[0]
    return _TotalLoss.apply(*terms)


def backward_total(loss: Tensor) -> None:
    """loss.backward() with a unit seed the loss Functions recognise (no ones_like fill, no seed * gradient launches)."""
    torch.autograd.backward(loss, _unit_seed(loss.device))


class _Distortion(torch.autograd.Function):
    """apply(weights, spacing_bins, mult) -> (metric, mult * metric): the distortion metric of get_metrics_dict and the loss
    term get_loss_dict makes of it, from one launch; the saved gradient is the TERM's (already scaled by mult)."""

    @staticmethod
    def forward(ctx, weights: Tensor, spacing_bins: Tensor, mult: float):
        R, n = weights.shape[0], weights.shape[1]
        w = _hip.require_device_tensor(weights.reshape(R, n), "weights")
        b = _hip.require_device_tensor(spacing_bins, "spacing_bins")
        pre = _take_precomputed(w.device, "dist", (_tensor_key(w), _tensor_key(b), float(mult)))
        loss, g = pre if pre is not None else _distortion_run(w, b, mult)
        ctx.g, ctx.shape, ctx.mult = g, weights.shape, mult
        ctx.set_materialize_grads(False)
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g_metric, g_term):
        out = None if g_term is None else _seeded(g_term, ctx.g)
        if g_metric is not None:  # someone differentiates the metric itself: its gradient is the term's / mult
            t = (g_metric / ctx.mult) * ctx.g
            out = t if out is None else out + t
        return (None if out is None else out.view(ctx.shape)), None, None


class _Interlevel(torch.autograd.Function):
    """Both proposal levels of NS interlevel_loss in one Function: apply(w, c, wp_0, cp_0, wp_1, cp_1, ...) — the levels'
    kernels add into one loss scalar, each already scaled by mult / (R n) (mult = the model's interlevel_loss_mult)."""

    @staticmethod
    def forward(ctx, mult: float, w: Tensor, c: Tensor, *levels: Tensor):
        R, n = w.shape[0], w.shape[1]
        w2 = _hip.require_device_tensor(w.reshape(R, n), "weights")
        c2 = _hip.require_device_tensor(c, "bins")
        pairs = []
        ctx.shapes = []
        for wp, cp in zip(levels[0::2], levels[1::2]):
            pairs.append((_hip.require_device_tensor(wp.reshape(R, wp.shape[1]), "proposal weights"),
                          _hip.require_device_tensor(cp, "proposal bins")))
            ctx.shapes.append(wp.shape)
        key = (_tensor_key(w2), _tensor_key(c2), float(mult)) + tuple((_tensor_key(wp2), _tensor_key(cp2)) for wp2, cp2 in pairs)
        pre = _take_precomputed(w2.device, "inter", key)
        # both levels in one launch (they are independent and each is latency-bound), each scaled by mult / (R n)
        loss, ctx.g = pre if pre is not None else _interlevel_run(mult, w2, c2, pairs)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        out = [None, None, None]
        for g, shape in zip(ctx.g, ctx.shapes):
            out += [_seeded(go, g).view(shape), None]
        return tuple(out)


class _ImageLosses(torch.autograd.Function):
    """rgb MSE, thermal MSE [REF thermal_nerf_model.py:294-295, 319-323] and the PSNR metric in one launch
    (tn_image_losses); apply(rgb [R,3], thermal [R,1], gt_rgb, gt_thermal) -> (rgb_loss, thermal_loss, psnr)."""

    @staticmethod
    def forward(ctx, rgb: Tensor, thermal: Tensor, gt_rgb: Tensor, gt_thermal: Tensor):
        R = rgb.shape[0]
        rgb_c, th_c = _hip.require_device_tensor(rgb, "rgb"), _hip.require_device_tensor(thermal.reshape(R), "thermal")
        g3, g1 = _hip.require_device_tensor(gt_rgb, "image"), _hip.require_device_tensor(gt_thermal.reshape(R), "thermal image")
        out, d_rgb, d_th = _f32((4,), rgb.device), _f32((R, 3), rgb.device), _f32((R, 1), rgb.device)
        _hip.check(_hip.load().tn_image_losses(rgb_c.data_ptr(), g3.data_ptr(), th_c.data_ptr(), g1.data_ptr(), R, out.data_ptr(),
                                               d_rgb.data_ptr(), d_th.data_ptr(), _stream()), "tn_image_losses")
        ctx.d_rgb, ctx.d_th = d_rgb, d_th
        ctx.set_materialize_grads(False)
        psnr = out[2]
        ctx.mark_non_differentiable(psnr)
        return out[0], out[1], psnr

    @staticmethod
    def backward(ctx, g_rgb, g_th, _g_psnr):
        return (None if g_rgb is None else _seeded(g_rgb, ctx.d_rgb), None if g_th is None else _seeded(g_th, ctx.d_th), None, None)


def image_losses(rgb: Tensor, thermal: Tensor, gt_rgb: Tensor, gt_thermal: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    return _ImageLosses.apply(rgb, thermal, gt_rgb, gt_thermal)


def distortion_loss(weights_list: Sequence[Tensor], ray_samples_list: Sequence, mult: Optional[float] = None) -> Tensor:
    """NS losses.distortion_loss (final level), O(S) per ray on the device.  With ``mult`` (non-zero) the returned metric carries
    ``.scaled_term = (mult, mult * metric)`` computed by the same launch, which get_loss_dict picks up instead of multiplying."""
    if not mult:
        return _Distortion.apply(weights_list[-1], ray_samples_list[-1].spacing_bins, 1.0)[0]
    metric, term = _Distortion.apply(weights_list[-1], ray_samples_list[-1].spacing_bins, float(mult))
    metric.scaled_term = (float(mult), term)
    return metric


def interlevel_loss(weights_list: Sequence[Tensor], ray_samples_list: Sequence, mult: float = 1.0) -> Tensor:
    """``mult`` * NS losses.interlevel_loss: final level detached, one term per proposal level."""
    c = ray_samples_list[-1].spacing_bins.detach()
    w = weights_list[-1].detach()
    levels = []
    for s, wp in zip(ray_samples_list[:-1], weights_list[:-1]):
        levels += [wp, s.spacing_bins]
    return _Interlevel.apply(float(mult), w, c, *levels)


# --------------------------------------------------------------------------------------------------
# model-facing entry
# --------------------------------------------------------------------------------------------------
def get_outputs_train(model, ray_bundle: RayBundle, jitter: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """Train-mode ``get_outputs`` with gradients [REF thermal_nerf_model.py:210-275]."""
    cfg = model.config
    if cfg.predict_normals:  # as the reference's own get_outputs ends (G9: tests/golden/predict_normals.json)
        from .fields import FieldHeadNames

        raise KeyError(FieldHeadNames.PRED_NORMALS)
    if cfg.num_proposal_iterations != 2:
        raise NotImplementedError("the training path implements two proposal iterations (the reference configuration)")
    # a fused optimizer may have stepped since the last forward without bumping parameter versions: the MFMA blob of the
    # fused taped forward (and every other derived copy) is rebuilt from the current weights
    model.invalidate_prepared()
    o = _hip.require_device_tensor(ray_bundle.origins, "origins")
    d = _hip.require_device_tensor(ray_bundle.directions, "directions")
    R, dev = o.shape[0], o.device
    nears = _hip.require_device_tensor(ray_bundle.nears.reshape(-1), "nears")
    fars = _hip.require_device_tensor(ray_bundle.fars.reshape(-1), "fars")
    if ray_bundle.camera_indices is None:
        raise AttributeError("Camera indices are not provided.")
    cam = _hip.require_device_tensor(ray_bundle.camera_indices.reshape(-1).to(torch.int32), "camera_indices", torch.int32)
    counts = (*cfg.num_proposal_samples_per_ray, cfg.num_nerf_samples_per_ray)
    if jitter is None:
        jitter = draw_jitter(R, counts, bool(cfg.use_single_jitter), dev)
    jitter = jitter_levels(jitter, R, counts, bool(cfg.use_single_jitter))[0]
    sampler = model.proposal_sampler
    updated = sampler._steps_since_update > sampler.update_sched(sampler._step) or sampler._step < 10
    params = model.named_parameter_lists()[1]
    (rgb, thermal, acc, w0, w1, w2, depth, expected, pd0, pd1, sp0, sp1, sp2, eu0, eu1, eu2) = RenderTrain.apply(
        model, o, d, nears, fars, cam, jitter, updated, *params)
    if updated:
        sampler._steps_since_update = 0
    return {
        "rgb": rgb, "accumulation": acc, "depth": depth, "expected_depth": expected,
        "weights_list": [w0, w1, w2],
        "ray_samples_list": [LazyRaySamples(ray_bundle, sp, eu, sampler.initial_sampler.uniform_spacing)
                             for sp, eu in ((sp0, eu0), (sp1, eu1), (sp2, eu2))],
        "prop_depth_0": pd0, "prop_depth_1": pd1, "thermal": thermal,
    }
