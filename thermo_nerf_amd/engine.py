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
        ``fuse_chunks`` (default): a frame of several chunks goes through ONE launch pair per ``launch_rays`` rays (as many whole
        chunks as fit ``max_workspace_bytes`` of bin-edge workspace) that keeps one depth-bound pair per chunk
        (tn_field_render_chunked_fwd): the reference's chunk-by-chunk result bit for bit, without under-filling the chip with
        1024-tile launches.  False, or a chunk that is not a multiple of 64 rays: one launch pair per chunk."""
        if model.training:
            raise RuntimeError("RayRenderEngine renders in eval mode; call model.eval() first")
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
        # rays per launch pair: whole chunks, as many as the workspace budget holds (4 (S+1 + 256 + 97) B per ray)
        self.launch_rays = self.chunk
        if self.fuse_chunks:
            per_ray = self.lib.tn_render_workspace_bytes(self.rc, self.chunk) / max(self.chunk, 1)
            self.launch_rays = max(1, int(max_workspace_bytes / max(per_ray, 1.0)) // self.chunk) * self.chunk
        # launches of 65 536 rays are 1024 waves — one per SIMD, half of what the field kernel needs to hide its gathers —
        # so consecutive launches go to alternating HIP streams (own workspace each) and overlap on the device
        # (default: 2 streams; 4 — the number of hardware queues HIP streams map onto — for small chunks)
        if streams is None:
            streams = 2 if self.launch_rays >= 32768 else 4
        self.num_streams = max(1, int(streams))
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
        if len(self._streams) < self.num_streams or (self._streams and self._streams[0].device != self._ws.device):
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.num_streams)]
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

    def _launch_pieces(self, start: int, end: int) -> List[Tuple[int, int]]:
        """[start, end) of a frame cut where the frame's launches are cut (multiples of ``launch_rays``)"""
        L = self.launch_rays
        return [(max(start, k * L), min(end, (k + 1) * L)) for k in range(start // L, (end - 1) // L + 1)] if end > start else []

    @torch.no_grad()
    def render(self, origins: Tensor, directions: Tensor, out: Optional[Dict[str, Tensor]] = None,
               record_events: bool = False, nears: Optional[Tensor] = None, fars: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """origins/directions [N,3] resident on the device -> dict of [N,C] tensors (keys = OUTPUT_KEYS).  ``nears`` / ``fars``
        [N] or [N,1]: per-ray planes already set on the bundle are honoured (NS SceneCollider.forward keeps them); absent,
        the model's NearFarCollider fills them.  ``expected_depth`` is clipped to the mid-point range of its CHUNK, exactly
        like the reference's per-chunk forward: it depends on ``chunk`` (the other outputs do not)."""
        o = _hip.require_device_tensor(origins, "origins")
        d = _hip.require_device_tensor(directions, "directions")
        n, dev = o.shape[0], o.device
        pieces = self._launch_pieces(0, n)
        self._buffers(dev, min(self.launch_rays, max(n, 1)), len(pieces))
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
        # launches overlapping on several streams fill the chip together: the lane = ray kernels pay from ~50 k rays in flight
        family = KERNEL_FAMILY[self.model.config.kernel_family]
        self.rc.kernel_family = 1 if (family == 0 and multi and self.launch_rays >= 49152) else family
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
            _hip.check(self.lib.tn_proposal_sample_fwd(prop0, prop1, self.rc, ins, outs, r, ws, wsn, stream),
                       "tn_proposal_sample_fwd")
            if record_events:
                e1.record(st)
            if chunked:  # one launch for the piece's chunks, one depth-bound pair per chunk, clipped per chunk
                _hip.check(self.lib.tn_field_render_chunked_fwd(fld, self.rc, ins, outs, r, ws, wsn, i, self.chunk,
                                                                bounds.data_ptr() + 8 * (i // self.chunk), 1, stream),
                           "tn_field_render_chunked_fwd")
            else:
                _hip.check(self.lib.tn_field_render_fwd(fld, self.rc, ins, outs, r, ws, wsn, stream), "tn_field_render_fwd")
            if record_events:
                e2.record(st)
                self.timings.append((e0, e1, e2))
        if multi:
            for st in self._streams:
                current.wait_stream(st)
        return out

    @torch.no_grad()
    def render_shard(self, origins: Tensor, directions: Tensor, start: int, frame_rays: int,
                     out: Optional[Dict[str, Tensor]] = None) -> Tuple[Dict[str, Tensor], Tensor]:
        """Rays [start, start + n) of a row-major frame of ``frame_rays`` rays whose reference chunking is this engine's
        ``chunk`` — a shard that need NOT begin or end on a chunk boundary, only on a multiple of 64 rays
        (distributed.render_frame_sharded_fine).  It is rendered by the launches ``render`` would use for that part of the frame,
        in the kernel form ``render`` would pick for the whole frame, so every per-ray output equals the unsharded frame's bit
        for bit; ``expected_depth`` is left UNCLIPPED and ``bounds[c]`` receives this shard's [min, max] of the sample mid-points
        in chunk ``c`` of the frame ((+inf, -inf) for chunks it does not touch).  The caller reduces ``bounds`` over the ranks
        (min / max) and calls ``apply_depth_bounds``.  Returns (outputs [n,C], bounds [chunks of the frame, 2])."""
        if not self.fuse_chunks:
            raise RuntimeError("render_shard needs fuse_chunks (a chunk size that is a multiple of 64 rays)")
        o = _hip.require_device_tensor(origins, "origins")
        d = _hip.require_device_tensor(directions, "directions")
        n, dev = o.shape[0], o.device
        if start < 0 or start + n > frame_rays or start % 64 != 0:
            raise ValueError("a shard lies inside the frame and starts on a multiple of 64 rays")
        pieces = self._launch_pieces(start, start + n)
        self._buffers(dev, max((j - i for i, j in pieces), default=1), len(pieces))
        if out is None:
            out = self.allocate_outputs(n, dev)
        bounds = torch.empty((-(-frame_rays // self.chunk), 2), dtype=torch.float32, device=dev)
        bounds[:, 0] = float("inf")
        bounds[:, 1] = float("-inf")
        if n == 0:
            return out, bounds
        prop0, prop1, fld = self.model._c_structs()
        ins = self._inputs(dev)
        ins.nears, ins.fars = self._nf[0].data_ptr(), self._nf[1].data_ptr()
        outs = _hip.tn_render_outputs()
        wsn = self._ws.shape[1]
        family = KERNEL_FAMILY[self.model.config.kernel_family]
        frame_launches = -(-frame_rays // self.launch_rays)
        multi_frame = self.num_streams > 1 and frame_launches > 1  # what render() of the WHOLE frame would decide
        fam_prop = fam_field = 1 if (family == 0 and multi_frame and self.launch_rays >= 49152) else family
        if fam_prop == 0:
            # "auto" is decided by the library from the size of the CALL (tn_render.hip: proposal pass lane = ray from 81 920
            # rays; tn_render_mfma.hip: field pass from 57 344): a shard runs the form the unsharded launch of the frame would
            # (tests/test_gpu_distributed.py renders frames on both sides of both thresholds)
            whole = min(self.launch_rays, frame_rays)
            fam_prop, fam_field = (1 if whole >= 81920 else 2), (1 if whole >= 57344 else 2)
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
            outs.rgb = out["rgb"].data_ptr() + 12 * i
            for k in OUTPUT_KEYS[1:]:
                setattr(outs, k, out[k].data_ptr() + 4 * i)
            self.rc.kernel_family = fam_prop
            _hip.check(self.lib.tn_proposal_sample_fwd(prop0, prop1, self.rc, ins, outs, r, ws, wsn, st.cuda_stream),
                       "tn_proposal_sample_fwd")
            self.rc.kernel_family = fam_field
            _hip.check(self.lib.tn_field_render_chunked_fwd(fld, self.rc, ins, outs, r, ws, wsn, p0, self.chunk,
                                                            bounds.data_ptr() + 8 * (p0 // self.chunk), 0, st.cuda_stream),
                       "tn_field_render_chunked_fwd")
        if multi:
            for st in self._streams:
                current.wait_stream(st)
        return out, bounds

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
