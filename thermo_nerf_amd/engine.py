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
    def __init__(self, model: ThermalNerfModel, chunk: Optional[int] = None, streams: Optional[int] = None) -> None:
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
        # a chunk of 65 536 rays is 1024 waves — one per SIMD, half of what the field kernel needs to hide its gathers —
        # so consecutive chunks go to alternating HIP streams (own workspace each) and overlap on the device
        # (default: 2 streams; 4 — the number of hardware queues HIP streams map onto — for small chunks)
        if streams is None:
            streams = 2 if self.chunk >= 32768 else 4
        self.num_streams = max(1, int(streams))
        self._streams: List[torch.cuda.Stream] = []
        self._ws: Optional[Tensor] = None
        self._nf: Optional[Tuple[Tensor, Tensor]] = None
        self._nf_key = None
        self.timings: List[Tuple[torch.cuda.Event, torch.cuda.Event, torch.cuda.Event]] = []

    def _buffers(self, dev) -> None:
        need = self.lib.tn_render_workspace_bytes(self.rc, self.chunk)
        need = (need + 255) // 256 * 256
        if self._ws is None or self._ws.shape[1] < need or self._ws.device != dev:
            self._ws = torch.empty((self.num_streams, need), dtype=torch.uint8, device=dev)
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.num_streams)]
        # NS NearFarCollider in eval: near plane reset to 0; keyed on the planes, so a collider edited after
        # the first render is picked up
        col = self.model.collider
        near = float(col.near_plane if not col.reset_near_plane else 0.0)
        key = (near, float(col.far_plane), str(dev), self.chunk)
        if self._nf is None or self._nf_key != key:
            self._nf = (torch.full((self.chunk,), near, dtype=torch.float32, device=dev),
                        torch.full((self.chunk,), float(col.far_plane), dtype=torch.float32, device=dev))
            self._nf_key = key

    def allocate_outputs(self, n: int, dev) -> Dict[str, Tensor]:
        out = {"rgb": torch.empty((n, 3), dtype=torch.float32, device=dev)}
        for k in OUTPUT_KEYS[1:]:
            out[k] = torch.empty((n, 1), dtype=torch.float32, device=dev)
        return out

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
        self._buffers(dev)
        if out is None:
            out = self.allocate_outputs(n, dev)
        prop0, prop1, fld = self.model._c_structs()
        ins = _hip.tn_render_inputs()
        if (nears is None) != (fars is None):
            raise ValueError("pass both nears and fars, or neither")
        if nears is not None:
            nears = _hip.require_device_tensor(nears.reshape(-1), "nears")
            fars = _hip.require_device_tensor(fars.reshape(-1), "fars")
            if nears.shape[0] != n or fars.shape[0] != n:
                raise ValueError("nears/fars must hold one value per ray")
        ins.camera_indices = None
        ins.jitter = None
        ins.lin_bins0 = linspace_bins(self.P0, dev).data_ptr()
        ins.u1 = pdf_positions(self.P1 + 1, dev, False).data_ptr()
        ins.u2 = pdf_positions(self.S + 1, dev, False).data_ptr()
        outs = _hip.tn_render_outputs()
        wsn = self._ws.shape[1]
        multi = self.num_streams > 1 and n > self.chunk
        # chunks overlapping on several streams fill the chip together: the lane = ray kernels pay from ~50 k rays in flight
        family = KERNEL_FAMILY[self.model.config.kernel_family]
        self.rc.kernel_family = 1 if (family == 0 and multi and self.chunk >= 49152) else family
        current = torch.cuda.current_stream(dev)
        if multi:
            for st in self._streams:
                st.wait_stream(current)  # inputs (and the prepared weights) were produced on the caller's stream
        for ci, i in enumerate(range(0, n, self.chunk)):
            r = min(self.chunk, n - i)
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
            _hip.check(self.lib.tn_field_render_fwd(fld, self.rc, ins, outs, r, ws, wsn, stream), "tn_field_render_fwd")
            if record_events:
                e2.record(st)
                self.timings.append((e0, e1, e2))
        if multi:
            for st in self._streams:
                current.wait_stream(st)
        return out

    def drain_timings(self) -> Tuple[List[float], List[float]]:
        """(proposal ms per launch, field ms per launch); call after a stream/device synchronise."""
        prop = [a.elapsed_time(b) for a, b, _ in self.timings]
        main = [b.elapsed_time(c) for _, b, c in self.timings]
        self.timings = []
        return prop, main
