"""RayRenderEngine — chunked eval-mode rendering of a resident ray set through the fused C-ABI entry points.

Counterpart of the reference's inference loop ``Renderer.render`` -> ``Model.get_outputs_for_camera_ray_bundle``
[REF thermo_nerf/render/renderer.py:160-201; SURVEY §8a a14], restructured for throughput: the ray set lives in
HBM, every chunk writes straight into slices of preallocated [N,C] outputs (no per-chunk ``cat``), all modalities
come out of ONE pass (the reference re-renders per modality, REF renderer.py:180-183), and the two kernels of a
chunk can be bracketed by HIP events on the launch stream for roofline accounting.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _hip
from .samplers import linspace_bins, pdf_positions
from .nerfacto_config.thermal_nerfacto import KERNEL_FAMILY
from .thermal_nerf.thermal_nerf_model import ThermalNerfModel

OUTPUT_KEYS = ("rgb", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1", "thermal")


class RayRenderEngine:
    def __init__(self, model: ThermalNerfModel, chunk: Optional[int] = None, streams: Optional[int] = None,
                 fuse_chunks: bool = True, max_workspace_bytes: int = 2 << 30) -> None:
        """``chunk``: the reference's ``eval_num_rays_per_chunk`` — the unit DepthRenderer("expected") clips over.
        ``fuse_chunks`` (default): a frame of several chunks goes through as few launch pairs as ``max_workspace_bytes`` allows,
        each keeping one depth-bound pair per chunk (tn_field_render_chunked_fwd): the reference's chunk-by-chunk result bit for
        bit, without under-filling the chip with 1024-tile launches.  False, or a chunk that is not a multiple of 64 rays: one
        launch pair per chunk.
        ``max_workspace_bytes`` bounds the engine's TOTAL bin-edge workspace (4 (S+1 + 256 + 97) B per ray in flight, all stream
        slots together; the peak is that or ONE chunk's workspace, whichever is larger): a frame that fits goes through one launch
        pair; a larger one is cut into k equal runs of whole chunks, k the smallest count whose run fits the budget divided by
        the number of streams (``frame_launch_rays``: 1080p at S=48 and the default 2 GiB -> 4 launches of 8 chunks, 1.6 GiB)."""
        if model.training:
            raise RuntimeError("RayRenderEngine renders in eval mode; call model.eval() first")
        if getattr(model.field, "staged", False):
            raise RuntimeError("RayRenderEngine drives the fused kernels (64-wide MLP layers); this field is staged (other widths): "
                               "render it through model.get_outputs_for_camera_ray_bundle")
        self.model = model
        self.chunk = int(chunk or model.config.eval_num_rays_per_chunk)
        self.lib = _hip.load()
        cfg = model.config
        self.P0, self.P1 = cfg.num_proposal_samples_per_ray
        self.S = cfg.num_nerf_samples_per_ray
        self.rc = _hip.tn_render_config()
        self.rc.num_proposal_samples[0], self.rc.num_proposal_samples[1] = self.P0, self.P1
        self.rc.num_nerf_samples = self.S
        self.rc.training = 0
        self.rc.pdf_anneal = float(model.proposal_sampler._anneal)
        self.rc.early_stop_transmittance = float(cfg.early_termination_eps)
        self.rc.kernel_family = 0
        self.rc.initial_sampler = int(model.proposal_sampler.initial_sampler.uniform_spacing)
        self.fuse_chunks = bool(fuse_chunks) and self.chunk % 64 == 0
        # launches of 65 536 rays are 1024 waves — one per SIMD, half of what the field kernel needs to hide its gathers —
        # so consecutive launches go to alternating HIP streams (own workspace each) and overlap on the device
        # (default: 2 streams; 4 — the number of hardware queues HIP streams map onto — for small chunks)
        if streams is None:
            streams = 2 if (self.fuse_chunks or self.chunk >= 32768) else 4
        self.num_streams = max(1, int(streams))
        # whole chunks per launch pair: what the whole budget holds (a frame that fits is ONE launch), and what one stream
        # slot's share of it holds (a longer frame: equal runs of at most that many chunks)
        self._chunks_whole = self._chunks_slot = 1
        if self.fuse_chunks:
            per_chunk = max(self.lib.tn_render_workspace_bytes(self.rc, self.chunk), 1)
            self._chunks_whole = max(1, int(max_workspace_bytes) // per_chunk)
            self._chunks_slot = max(1, int(max_workspace_bytes) // self.num_streams // per_chunk)
        self.launch_rays = self._chunks_slot * self.chunk  # (the largest launch of a frame that does not fit one)
        self._streams: List[torch.cuda.Stream] = []
        self._ws: Optional[Tensor] = None
        self._ws_rays = 0
        self._nf: Optional[Tuple[Tensor, Tensor]] = None
        self._nf_key = None
        self.timings: List[Tuple[torch.cuda.Event, torch.cuda.Event, torch.cuda.Event]] = []

    def _buffers(self, dev, rays_per_launch: Optional[int] = None, launches: int = 2) -> None:
        rays = int(rays_per_launch or self.launch_rays)
        need = self.lib.tn_render_workspace_bytes(self.rc, rays)
        need = (need + 255) // 256 * 256
        slots = min(self.num_streams, max(launches, 1))
        if self._ws is None or self._ws.shape[1] < need or self._ws.shape[0] < slots or self._ws.device != dev:
            self._ws = None  # (release before growing: two generations of a GB-sized workspace need not coexist)
            self._ws = torch.empty((slots, need), dtype=torch.uint8, device=dev)
        if slots > 1 and (len(self._streams) < self.num_streams or self._streams[0].device != self._ws.device):
            # (only a frame of several launches needs them) streams on distinct hardware queues: two pool streams may share one
            # and run their launches back to back
            self._streams = _hip.concurrent_streams(dev, self.num_streams, with_main=False)
        self._ws_rays = max(self._ws_rays, rays)
        # NS NearFarCollider in eval: near plane reset to 0; keyed on the planes, so a collider edited after
        # the first render is picked up
        col = self.model.collider
        near = float(col.near_plane if not col.reset_near_plane else 0.0)
        if self._nf is not None and self._nf[0].shape[0] >= rays and self._nf_key == (near, float(col.far_plane), str(dev)):
            return
        self._nf = (torch.full((rays,), near, dtype=torch.float32, device=dev),
                    torch.full((rays,), float(col.far_plane), dtype=torch.float32, device=dev))
        self._nf_key = (near, float(col.far_plane), str(dev))

    def allocate_outputs(self, n: int, dev) -> Dict[str, Tensor]:
        out = {"rgb": torch.empty((n, 3), dtype=torch.float32, device=dev)}
        for k in OUTPUT_KEYS[1:]:
            out[k] = torch.empty((n, 1), dtype=torch.float32, device=dev)
        return out

    def _inputs(self, dev):
        ins = _hip.tn_render_inputs()
        ins.camera_indices = None
        ins.jitter = None
        ins.lin_bins0 = linspace_bins(self.P0, dev).data_ptr()
        ins.u1 = pdf_positions(self.P1 + 1, dev, False).data_ptr()
        ins.u2 = pdf_positions(self.S + 1, dev, False).data_ptr()
        return ins

    def frame_launch_rays(self, frame_rays: int) -> int:
        """Rays per launch pair of a ``frame_rays``-ray frame: the frame if its workspace fits the budget, else equal runs of whole
        chunks (k = the fewest launches whose run fits one stream slot's share of the budget)."""
        n_chunks = -(-max(int(frame_rays), 1) // self.chunk)
        if n_chunks <= self._chunks_whole:
            return n_chunks * self.chunk
        k = -(-n_chunks // self._chunks_slot)
        return -(-n_chunks // k) * self.chunk

    def _launch_pieces(self, start: int, end: int, frame_rays: int) -> List[Tuple[int, int]]:
        """[start, end) of a frame cut where the frame's launches are cut (multiples of ``frame_launch_rays``)"""
        L = self.frame_launch_rays(frame_rays)
        return [(max(start, k * L), min(end, (k + 1) * L)) for k in range(start // L, (end - 1) // L + 1)] if end > start else []

    def _forms(self, fld, frame_rays: int, piece_start: int, sample_split: Optional[int] = None) -> Tuple[int, int, int]:
        """(proposal form, field form, sample segments per tile) of the frame launch that holds ray ``piece_start`` — what ``render``
        of the WHOLE frame runs there; the decisions are the library's (tn_render_kernel_form, tn_render_sample_split), never
        re-derived here.  ``sample_split``: None = the library's choice for a call of the frame's size, k = forced (capped by the library)."""
        L = self.frame_launch_rays(frame_rays)
        family = KERNEL_FAMILY[self.model.config.kernel_family]
        k = piece_start // L
        whole = min((k + 1) * L, frame_rays) - k * L
        # STATELESS: both library queries see the caller's REQUEST (0 = the library decides, k = forced) — never the split `render`
        # left in self.rc for the previous piece's launch (piece 0 was decided with 0, later pieces with the frame's
        # split, which moves the field's lane = ray threshold from 8 192 to 57 344 rays when it is 1); self.rc is restored on exit
        request = self._split_request(sample_split)
        saved = self.rc.kernel_family, self.rc.sample_split
        try:
            self.rc.sample_split = request
            if family == 0 and self.num_streams > 1 and frame_rays > L and L >= 49152:
                # launches overlapping on several streams fill the chip together: the lane = ray kernels pay from ~50 k rays in flight
                prop = field = 1
            else:
                self.rc.kernel_family = family
                prop, field = (int(self.lib.tn_render_kernel_form(None, self.rc, whole, 0)),
                               int(self.lib.tn_render_kernel_form(fld, self.rc, whole, 1)))
            # segments per tile: ONE value per frame, whatever launches it is cut into (a ray's bits must not depend on the launch
            # that holds it, nor on the number of streams): the library's choice for a call of the WHOLE frame's size — 1 for a frame
            # of 400 k rays or more, several for the small frames whose tiles would leave most wave slots idle
            self.rc.kernel_family = field
            split = int(self.lib.tn_render_sample_split(fld, self.rc, frame_rays))
        finally:
            self.rc.kernel_family, self.rc.sample_split = saved
        return prop, field, split

    def _split_request(self, sample_split: Optional[int]) -> int:
        """tn_render_config.sample_split for a call: the per-call argument, else the model's config.sample_split (0 = the library
        decides, 1 = never: the serial march's bits, as model.get_outputs honours it), else 0"""
        if sample_split is None:
            sample_split = int(getattr(self.model.config, "sample_split", 0) or 0)
            return max(sample_split, 0)
        return max(int(sample_split), 1)

    @torch.no_grad()
    def render(self, origins: Tensor, directions: Tensor, out: Optional[Dict[str, Tensor]] = None,
               record_events: bool = False, nears: Optional[Tensor] = None, fars: Optional[Tensor] = None,
               sample_split: Optional[int] = None) -> Dict[str, Tensor]:
        """origins/directions [N,3] resident on the device -> dict of [N,C] tensors (keys = OUTPUT_KEYS).  ``nears`` / ``fars``
        [N] or [N,1]: per-ray planes already set on the bundle are honoured (NS SceneCollider.forward keeps them); absent,
        the model's NearFarCollider fills them.  ``expected_depth`` is clipped to the mid-point range of its CHUNK, exactly
        like the reference's per-chunk forward: it depends on ``chunk`` (the other outputs do not).
        ``sample_split``: segments per 64-ray tile of the field pass (tn_render_config.sample_split): None = the library's choice
        for a call of this frame's size (1 from ~500 k rays up; a small frame is marched in shorter pieces on more waves), k = that
        many.  One value per frame, whatever launches it is cut into."""
        o = _hip.require_device_tensor(origins, "origins")
        d = _hip.require_device_tensor(directions, "directions")
        n, dev = o.shape[0], o.device
        pieces = self._launch_pieces(0, n, n)
        self._buffers(dev, max((j - i for i, j in pieces), default=1), len(pieces))
        if out is None:
            out = self.allocate_outputs(n, dev)
        prop0, prop1, fld = self.model._c_structs()
        ins = self._inputs(dev)
        if (nears is None) != (fars is None):
            raise ValueError("pass both nears and fars, or neither")
        if nears is not None:
            nears = _hip.require_device_tensor(nears.reshape(-1), "nears")
            fars = _hip.require_device_tensor(fars.reshape(-1), "fars")
            if nears.shape[0] != n or fars.shape[0] != n:
                raise ValueError("nears/fars must hold one value per ray")
        outs = _hip.tn_render_outputs()
        wsn = self._ws.shape[1]
        multi = self.num_streams > 1 and len(pieces) > 1
        family = KERNEL_FAMILY[self.model.config.kernel_family]
        chunked = self.fuse_chunks and n > self.chunk
        bounds = torch.empty((-(-n // self.chunk), 2), dtype=torch.float32, device=dev) if chunked else None
        current = torch.cuda.current_stream(dev)
        if multi:
            for st in self._streams:
                st.wait_stream(current)  # inputs (and the prepared weights) were produced on the caller's stream
        for ci, (i, j) in enumerate(pieces):
            r = j - i
            slot = ci % self.num_streams if multi else 0
            st = self._streams[slot] if multi else current
            stream = st.cuda_stream
            ws = self._ws[slot].data_ptr()
            ins.origins, ins.directions = o.data_ptr() + 12 * i, d.data_ptr() + 12 * i
            if nears is not None:
                ins.nears, ins.fars = nears.data_ptr() + 4 * i, fars.data_ptr() + 4 * i
            else:
                ins.nears, ins.fars = self._nf[0].data_ptr(), self._nf[1].data_ptr()
            outs.rgb = out["rgb"].data_ptr() + 12 * i
            for k in OUTPUT_KEYS[1:]:
                setattr(outs, k, out[k].data_ptr() + 4 * i)
            if record_events:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record(st)
            form_prop, form_field, split = self._forms(fld, n, i, sample_split)
            self.rc.kernel_family, self.rc.sample_split = form_prop, split
            _hip.check(self.lib.tn_proposal_sample_fwd(prop0, prop1, self.rc, ins, outs, r, ws, wsn, stream),
                       "tn_proposal_sample_fwd")
            if record_events:
                e1.record(st)
            self.rc.kernel_family = form_field
            if chunked:  # one launch for the piece's chunks, one depth-bound pair per chunk, clipped per chunk
                _hip.check(self.lib.tn_field_render_chunked_fwd(fld, self.rc, ins, outs, r, ws, wsn, i, self.chunk,
                                                                bounds.data_ptr() + 8 * (i // self.chunk), 1, stream),
                           "tn_field_render_chunked_fwd")
            else:
                _hip.check(self.lib.tn_field_render_fwd(fld, self.rc, ins, outs, r, ws, wsn, stream), "tn_field_render_fwd")
            if record_events:
                e2.record(st)
                self.timings.append((e0, e1, e2))
        self.rc.kernel_family, self.rc.sample_split = family, 0
        if multi:
            for st in self._streams:
                current.wait_stream(st)
        return out

    @torch.no_grad()
    def render_shard(self, origins: Tensor, directions: Tensor, start: int, frame_rays: int,
                     out: Optional[Dict[str, Tensor]] = None, nears: Optional[Tensor] = None,
                     fars: Optional[Tensor] = None, sample_split: Optional[int] = None) -> Tuple[Dict[str, Tensor], Tensor]:
        """Rays [start, start + n) of a row-major frame of ``frame_rays`` rays whose reference chunking is this engine's
        ``chunk`` — a shard that need NOT begin or end on a chunk boundary, only on a multiple of 64 rays
        (distributed.render_frame_sharded_fine).  It is rendered by the launches ``render`` would use for that part of the frame,
        in the kernel form ``render`` would pick for the whole frame, so every per-ray output equals the unsharded frame's bit
        for bit; ``expected_depth`` is left UNCLIPPED and ``bounds[c]`` receives this shard's [min, max] of the sample mid-points
        in chunk ``c`` of the frame ((+inf, -inf) for chunks it does not touch).  The caller reduces ``bounds`` over the ranks
        (min / max) and calls ``apply_depth_bounds``.  ``nears`` / ``fars`` [n] or [n,1]: the shard's slice of per-ray planes a
        bundle already carries (as in ``render``); absent, the collider's.  An EMPTY shard (more ranks than 64-ray tiles) is valid
        wherever it starts.  ``sample_split``: as in ``render`` — None = what the unsharded frame's launches use (1 for a frame: the
        shard, however small, then marches whole tiles); k = k segments per tile, bit-equal to ``render(frame, sample_split=k)``
        (``shard_sample_split`` proposes the k that suits a shard's size).  Returns (outputs [n,C], bounds [chunks of the frame, 2])."""
        if not self.fuse_chunks:
            raise RuntimeError("render_shard needs fuse_chunks (a chunk size that is a multiple of 64 rays)")
        o = _hip.require_device_tensor(origins, "origins")
        d = _hip.require_device_tensor(directions, "directions")
        n, dev = o.shape[0], o.device
        if out is None:
            out = self.allocate_outputs(n, dev)
        bounds = torch.empty((-(-frame_rays // self.chunk), 2), dtype=torch.float32, device=dev)
        bounds[:, 0] = float("inf")
        bounds[:, 1] = float("-inf")
        if n == 0:  # (before the alignment check: distributed.ray_block hands a surplus rank [frame_rays, frame_rays))
            return out, bounds
        if start < 0 or start + n > frame_rays or start % 64 != 0:
            raise ValueError("a shard lies inside the frame and starts on a multiple of 64 rays")
        if (nears is None) != (fars is None):
            raise ValueError("pass both nears and fars, or neither")
        if nears is not None:
            nears = _hip.require_device_tensor(nears.reshape(-1), "nears")
            fars = _hip.require_device_tensor(fars.reshape(-1), "fars")
            if nears.shape[0] != n or fars.shape[0] != n:
                raise ValueError("nears/fars must hold one value per ray of the shard")
        pieces = self._launch_pieces(start, start + n, frame_rays)
        self._buffers(dev, max((j - i for i, j in pieces), default=1), len(pieces))
        prop0, prop1, fld = self.model._c_structs()
        ins = self._inputs(dev)
        ins.nears, ins.fars = self._nf[0].data_ptr(), self._nf[1].data_ptr()
        outs = _hip.tn_render_outputs()
        wsn = self._ws.shape[1]
        family = KERNEL_FAMILY[self.model.config.kernel_family]
        current = torch.cuda.current_stream(dev)
        multi = self.num_streams > 1 and len(pieces) > 1
        if multi:
            for st in self._streams:
                st.wait_stream(current)
        for ci, (p0, p1) in enumerate(pieces):
            i, r = p0 - start, p1 - p0
            slot = ci % self.num_streams if multi else 0
            st = self._streams[slot] if multi else current
            ws = self._ws[slot].data_ptr()
            ins.origins, ins.directions = o.data_ptr() + 12 * i, d.data_ptr() + 12 * i
            if nears is not None:
                ins.nears, ins.fars = nears.data_ptr() + 4 * i, fars.data_ptr() + 4 * i
            outs.rgb = out["rgb"].data_ptr() + 12 * i
            for k in OUTPUT_KEYS[1:]:
                setattr(outs, k, out[k].data_ptr() + 4 * i)
            # the forms the unsharded frame's launch over these rays runs in (frames on both sides of every threshold, both
            # precisions: tests/test_gpu_distributed.py)
            fam_prop, fam_field, split = self._forms(fld, frame_rays, p0, sample_split)
            self.rc.kernel_family, self.rc.sample_split = fam_prop, split
            _hip.check(self.lib.tn_proposal_sample_fwd(prop0, prop1, self.rc, ins, outs, r, ws, wsn, st.cuda_stream),
                       "tn_proposal_sample_fwd")
            self.rc.kernel_family = fam_field
            _hip.check(self.lib.tn_field_render_chunked_fwd(fld, self.rc, ins, outs, r, ws, wsn, p0, self.chunk,
                                                            bounds.data_ptr() + 8 * (p0 // self.chunk), 0, st.cuda_stream),
                       "tn_field_render_chunked_fwd")
        self.rc.kernel_family, self.rc.sample_split = family, 0  # (the per-launch forms above are not the engine's setting)
        if multi:
            for st in self._streams:
                current.wait_stream(st)
        return out, bounds

    def shard_sample_split(self, shard_rays: int) -> int:
        """The segments per tile the library would pick for a lane = ray call of ``shard_rays`` rays (tn_render_sample_split): what a
        caller that cuts a frame into shards of that size passes as ``sample_split`` to ``render_shard`` — and to ``render`` for the
        unsharded frame it compares with: the split is a property of the frame, the same on every rank."""
        _, _, fld = self.model._c_structs()
        saved = self.rc.kernel_family, self.rc.sample_split
        self.rc.kernel_family, self.rc.sample_split = 1, 0
        k = int(self.lib.tn_render_sample_split(fld, self.rc, max(int(shard_rays), 1)))
        self.rc.kernel_family, self.rc.sample_split = saved
        return k

    def apply_depth_bounds(self, out: Dict[str, Tensor], start: int, bounds: Tensor) -> None:
        """the expected-depth clip of a ``render_shard`` result with the per-chunk bounds reduced over all ranks"""
        b = _hip.require_device_tensor(bounds, "bounds")
        n = out["expected_depth"].shape[0]
        _hip.check(self.lib.tn_expected_depth_clip_chunked(out["expected_depth"].data_ptr(), n, start, self.chunk,
                                                           b.data_ptr() + 8 * (start // self.chunk), _hip.current_stream()),
                   "tn_expected_depth_clip_chunked")

    def drain_timings(self) -> Tuple[List[float], List[float]]:
        """(proposal ms per launch, field ms per launch); call after a stream/device synchronise."""
        prop = [a.elapsed_time(b) for a, b, _ in self.timings]
        main = [b.elapsed_time(c) for _, b, c in self.timings]
        self.timings = []
        return prop, main
