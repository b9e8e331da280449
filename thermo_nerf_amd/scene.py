"""Scene-level helpers with nerfstudio's names: SceneBox, SceneContraction, NearFarCollider
(imported by the reference at [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:8,10,27])."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor, nn

from .rays import RayBundle


@dataclass
class SceneBox:
    """NS SceneBox: aabb [2,3] (min, max).  The thermal dataparser uses +-1 [REF thermal_dataparser.py:242-251]."""

    aabb: Tensor

    @staticmethod
    def unit(scale: float = 1.0) -> "SceneBox":
        return SceneBox(torch.tensor([[-scale] * 3, [scale] * 3], dtype=torch.float32))


class SceneContraction(nn.Module):
    """NS SceneContraction marker.  The contraction itself (L-inf) runs inside the HIP field kernels;
    only ``order=inf`` — the one built at [REF thermal_nerf_model.py:94] — is implemented."""

    def __init__(self, order=float("inf")) -> None:
        super().__init__()
        if order != float("inf"):
            raise NotImplementedError("only SceneContraction(order=inf) is implemented (REF thermal_nerf_model.py:94)")
        self.order = order


class NearFarCollider(nn.Module):
    """NS NearFarCollider: eval resets the near plane to 0 when reset_near_plane.  The two constant planes it
    attaches are SHARED between calls of the same shape (a training loop asks for the same two tensors every step): treat
    ``ray_bundle.nears`` / ``.fars`` as read-only, or clone them before editing in place."""

    def __init__(self, near_plane: float, far_plane: float, reset_near_plane: bool = True) -> None:
        super().__init__()
        self.near_plane = near_plane
        self.far_plane = far_plane
        self.reset_near_plane = reset_near_plane
        self._last = None  # (key, nears, fars): the two constant planes of the previous call, reused when nothing changed

    def set_nears_and_fars(self, ray_bundle: RayBundle) -> RayBundle:
        near = self.near_plane if (self.training or not self.reset_near_plane) else 0.0
        shape = (*ray_bundle.origins.shape[:-1], 1)
        key = (shape, float(near), float(self.far_plane), ray_bundle.origins.device)
        if self._last is None or self._last[0] != key:
            # constant tensors, read-only downstream: a training loop asks for the same two every step
            self._last = (key, torch.full(shape, float(near), dtype=torch.float32, device=ray_bundle.origins.device),
                          torch.full(shape, float(self.far_plane), dtype=torch.float32, device=ray_bundle.origins.device))
        ray_bundle.nears, ray_bundle.fars = self._last[1], self._last[2]
        return ray_bundle

    def forward(self, ray_bundle: RayBundle) -> RayBundle:
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            return ray_bundle
        return self.set_nears_and_fars(ray_bundle)
