"""Ray containers with the nerfstudio names the reference's hot path consumes
(``nerfstudio.cameras.rays``: RayBundle, RaySamples, Frustums — imported at
[REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:7]).

Only the members the ThermoNeRF path touches exist.  Arithmetic (positions, weights) is done by the HIP
kernels behind ``thermo_nerf_amd._hip``; these classes only carry device tensors.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from . import _hip


@dataclass
class Frustums:
    """origins/directions [R,n,3] (broadcast views), starts/ends [R,n,1], pixel_area [R,n,1]."""

    origins: Tensor
    directions: Tensor
    starts: Tensor
    ends: Tensor
    pixel_area: Optional[Tensor] = None

    @property
    def shape(self):
        return self.starts.shape[:-1]

    def get_positions(self) -> Tensor:
        """NS Frustums.get_positions: origins + directions * (starts + ends) / 2 -> [R,n,3]."""
        R, n = self.starts.shape[0], self.starts.shape[1]
        o = _hip.require_device_tensor(self.origins[:, 0, :], "origins")
        d = _hip.require_device_tensor(self.directions[:, 0, :], "directions")
        s = _hip.require_device_tensor(self.starts[..., 0], "starts")
        e = _hip.require_device_tensor(self.ends[..., 0], "ends")
        pos = torch.empty((R, n, 3), dtype=torch.float32, device=s.device)
        lib = _hip.load()
        _hip.check(
            lib.tn_frustum_positions(o.data_ptr(), d.data_ptr(), s.data_ptr(), e.data_ptr(), R, n, pos.data_ptr(),
                                     _hip.current_stream()),
            "tn_frustum_positions",
        )
        return pos


@dataclass
class RaySamples:
    """NS RaySamples (non-packed): all sample tensors are [R,n,1]."""

    frustums: Frustums
    camera_indices: Optional[Tensor] = None
    deltas: Optional[Tensor] = None
    spacing_starts: Optional[Tensor] = None
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, Tensor]] = None
    # [R,n+1] bin edges kept alongside (spacing / euclidean); what the HIP samplers exchange
    spacing_bins: Optional[Tensor] = None
    eucl_bins: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    # which SpacedSampler made the bins: False = UniformLinDispPiecewiseSampler, True = UniformSampler
    uniform_spacing: bool = False

    @property
    def shape(self):
        return self.frustums.shape

    def get_weights(self, densities: Tensor) -> Tensor:
        """NS RaySamples.get_weights, called at [REF thermal_nerf_model.py:233]: densities [R,n,1] -> [R,n,1]."""
        R, n = densities.shape[0], densities.shape[1]
        dl = _hip.require_device_tensor(self.deltas[..., 0], "deltas")
        dn = _hip.require_device_tensor(densities[..., 0], "densities")
        w = torch.empty((R, n), dtype=torch.float32, device=dn.device)
        lib = _hip.load()
        _hip.check(lib.tn_weights_fwd(dl.data_ptr(), dn.data_ptr(), R, n, w.data_ptr(), _hip.current_stream()),
                   "tn_weights_fwd")
        return w[..., None]


@dataclass
class RayBundle:
    """NS RayBundle: origins/directions [*bs,3], pixel_area [*bs,1], camera_indices [*bs,1] int, nears/fars [*bs,1]."""

    origins: Tensor
    directions: Tensor
    pixel_area: Optional[Tensor] = None
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def __len__(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n

    def _map(self, fn) -> "RayBundle":
        return RayBundle(
            origins=fn(self.origins), directions=fn(self.directions),
            pixel_area=None if self.pixel_area is None else fn(self.pixel_area),
            camera_indices=None if self.camera_indices is None else fn(self.camera_indices),
            nears=None if self.nears is None else fn(self.nears),
            fars=None if self.fars is None else fn(self.fars),
            metadata={k: fn(v) for k, v in self.metadata.items()},
            times=None if self.times is None else fn(self.times),
        )

    def flatten(self) -> "RayBundle":
        return self._map(lambda t: t.reshape(-1, t.shape[-1]))

    def to(self, device) -> "RayBundle":
        return self._map(lambda t: t.to(device))

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        """NS RayBundle.get_row_major_sliced_ray_bundle (used by Model.get_outputs_for_camera_ray_bundle)."""
        return self._map(lambda t: t.reshape(-1, t.shape[-1])[start_idx:end_idx])

    def get_ray_samples(self, bin_starts: Tensor, bin_ends: Tensor, spacing_starts: Optional[Tensor] = None,
                        spacing_ends: Optional[Tensor] = None,
                        spacing_to_euclidean_fn: Optional[Callable] = None) -> RaySamples:
        """NS RayBundle.get_ray_samples: bin_* are [R,n,1]."""
        n = bin_starts.shape[-2]
        deltas = bin_ends - bin_starts
        cam = None if self.camera_indices is None else self.camera_indices[..., None, :].expand(-1, n, -1)
        fr = Frustums(
            origins=self.origins[..., None, :].expand(-1, n, -1),
            directions=self.directions[..., None, :].expand(-1, n, -1),
            starts=bin_starts, ends=bin_ends,
            pixel_area=None if self.pixel_area is None else self.pixel_area[..., None, :].expand(-1, n, -1),
        )
        return RaySamples(frustums=fr, camera_indices=cam, deltas=deltas, spacing_starts=spacing_starts,
                          spacing_ends=spacing_ends, spacing_to_euclidean_fn=spacing_to_euclidean_fn,
                          nears=self.nears, fars=self.fars)
