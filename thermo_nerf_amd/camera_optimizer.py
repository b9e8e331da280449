"""Per-camera pose refinement holder with nerfstudio's parameter name ``pose_adjustment`` [N,6]
(NS CameraOptimizer; configured at [REF thermo_nerf/nerfacto_config/thermal_nerfacto.py:38-40], applied in
training at [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:218-219]).  Training-side glue: a few tiny
torch ops per batch on [R,3] tensors, outside the rendering hot path."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal

import torch
from torch import Tensor, nn

from .rays import RayBundle


@dataclass
class CameraOptimizerConfig:
    mode: Literal["off", "SO3xR3", "SE3"] = "off"

    def setup(self, num_cameras: int, device="cpu") -> "CameraOptimizer":
        return CameraOptimizer(self, num_cameras, device)


def _so3xr3_exp(tangent: Tensor) -> Tensor:
    """Rodrigues rotation from the last 3 components + plain translation from the first 3 -> [N,3,4]."""
    t, w = tangent[:, :3], tangent[:, 3:]
    theta = torch.clamp((w * w).sum(dim=1), min=1e-4).sqrt()
    a = torch.sin(theta) / theta
    b = (1.0 - torch.cos(theta)) / (theta * theta)
    K = torch.zeros((tangent.shape[0], 3, 3), dtype=tangent.dtype, device=tangent.device)
    K[:, 0, 1], K[:, 0, 2] = -w[:, 2], w[:, 1]
    K[:, 1, 0], K[:, 1, 2] = w[:, 2], -w[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -w[:, 1], w[:, 0]
    rot = a[:, None, None] * K + b[:, None, None] * torch.bmm(K, K) + torch.eye(3, dtype=tangent.dtype, device=tangent.device)
    return torch.cat([rot, t[:, :, None]], dim=2)


class CameraOptimizer(nn.Module):
    def __init__(self, config: CameraOptimizerConfig, num_cameras: int, device="cpu") -> None:
        super().__init__()
        self.config = config
        self.num_cameras = num_cameras
        if config.mode == "SE3":
            raise NotImplementedError('camera_optimizer_mode "SE3" is not implemented (reference recommends SO3xR3)')
        if config.mode != "off":
            self.pose_adjustment = nn.Parameter(torch.zeros((num_cameras, 6), device=device))

    def forward(self, indices: Tensor) -> Tensor:
        if self.config.mode == "off":
            eye = torch.eye(4, device=indices.device)[None, :3, :4]
            return eye.repeat(indices.shape[0], 1, 1)
        # exponentiate once per CAMERA, then gather per ray (index_select: its backward is an index_add, not the
        # serial indexing_backward kernel of advanced indexing)
        return _so3xr3_exp(self.pose_adjustment).index_select(0, indices)

    def apply_to_raybundle(self, raybundle: RayBundle) -> None:
        if self.config.mode == "off":
            return
        m = self(raybundle.camera_indices.squeeze(-1).long())
        raybundle.origins = raybundle.origins + m[:, :3, 3]
        raybundle.directions = torch.bmm(m[:, :3, :3], raybundle.directions[..., None]).squeeze(-1)
