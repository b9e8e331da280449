"""Per-camera pose refinement holder with nerfstudio's parameter name ``pose_adjustment`` [N,6]
(NS CameraOptimizer; configured at [REF thermo_nerf/nerfacto_config/thermal_nerfacto.py:38-40], applied in
training at [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:218-219]).  ``apply_to_raybundle`` — the first statement
of the training forward — is one HIP kernel each way (tn_camera_opt_fwd / tn_camera_opt_bwd); ``forward(indices)`` (the
[N,3,4] correction matrices, which nothing on the rendering path asks for) stays a handful of torch ops."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal

import torch
from torch import Tensor, nn

from . import _hip
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


def _se3_translation(tangent: Tensor) -> Tensor:
    """NS cameras/lie_groups.exp_map_SE3: the translation of exp([u | w]) is V(w) u, V = I + b K + c K^2 with
    b = (1 - cos t) / t^2, c = (t - sin t) / t^3, t^2 = max(|w|^2, 1e-4) — written with cross products (K u = w x u).
    Returns the [N,6] table [V(w) u | w]: the SO3xR3 form of the same rigid motions, which the HIP kernel applies per ray."""
    u, w = tangent[:, :3], tangent[:, 3:]
    theta = torch.clamp((w * w).sum(dim=1), min=1e-4).sqrt()
    b = ((1.0 - torch.cos(theta)) / (theta * theta))[:, None]
    c = ((theta - torch.sin(theta)) / (theta * theta * theta))[:, None]
    wu = torch.cross(w, u, dim=1)
    return torch.cat([u + b * wu + c * torch.cross(w, wu, dim=1), w], dim=1)


def _se3_exp(tangent: Tensor) -> Tensor:
    return _so3xr3_exp(_se3_translation(tangent))


class _ApplyPoseAdjustment(torch.autograd.Function):
    """origins + t_c, R(w_c) directions with c = the ray's camera (NS exp_map_SO3xR3 + apply_to_raybundle) on the device."""

    @staticmethod
    def forward(ctx, pose: Tensor, cam: Tensor, origins: Tensor, directions: Tensor):
        pose = _hip.require_device_tensor(pose.detach(), "pose_adjustment")
        cam = _hip.require_device_tensor(cam, "camera_indices", dtype=torch.int64)
        o = _hip.require_device_tensor(origins.detach(), "origins")
        d = _hip.require_device_tensor(directions.detach(), "directions")
        out_o, out_d = torch.empty_like(o), torch.empty_like(d)
        _hip.check(_hip.load().tn_camera_opt_fwd(pose.data_ptr(), cam.data_ptr(), o.data_ptr(), d.data_ptr(), o.shape[0],
                                                 pose.shape[0], out_o.data_ptr(), out_d.data_ptr(), _hip.current_stream()),
                   "tn_camera_opt_fwd")
        ctx.save_for_backward(pose, cam, d)
        ctx.set_materialize_grads(False)
        return out_o, out_d

    @staticmethod
    def backward(ctx, g_o, g_d):
        pose, cam, d = ctx.saved_tensors
        if g_o is None and g_d is None:
            return None, None, None, None
        g_o = None if g_o is None else _hip.require_device_tensor(g_o, "d origins")
        g_d = None if g_d is None else _hip.require_device_tensor(g_d, "d directions")
        d_pose = _hip.fresh_zeros(tuple(pose.shape), pose.device)
        d_dirs = torch.empty_like(d) if ctx.needs_input_grad[3] else None
        _hip.check(_hip.load().tn_camera_opt_bwd(pose.data_ptr(), cam.data_ptr(), d.data_ptr(), _hip.ptr(g_o), _hip.ptr(g_d),
                                                 d.shape[0], pose.shape[0], d_pose.data_ptr(), _hip.ptr(d_dirs),
                                                 _hip.current_stream()), "tn_camera_opt_bwd")
        if d_dirs is not None and g_d is None:
            d_dirs.zero_()
        return d_pose, None, (g_o if ctx.needs_input_grad[2] else None), d_dirs


class CameraOptimizer(nn.Module):
    def __init__(self, config: CameraOptimizerConfig, num_cameras: int, device="cpu") -> None:
        super().__init__()
        self.config = config
        self.num_cameras = num_cameras
        if config.mode not in ("off", "SO3xR3", "SE3"):
            raise ValueError(f"unknown camera optimizer mode {config.mode!r}")
        if config.mode != "off":
            self.pose_adjustment = nn.Parameter(torch.zeros((num_cameras, 6), device=device))

    def forward(self, indices: Tensor) -> Tensor:
        if self.config.mode == "off":
            eye = torch.eye(4, device=indices.device)[None, :3, :4]
            return eye.repeat(indices.shape[0], 1, 1)
        # exponentiate once per CAMERA, then gather per ray (index_select: its backward is an index_add, not the
        # serial indexing_backward kernel of advanced indexing)
        exp = _se3_exp if self.config.mode == "SE3" else _so3xr3_exp
        return exp(self.pose_adjustment).index_select(0, indices)

    def apply_to_raybundle(self, raybundle: RayBundle) -> None:
        if self.config.mode == "off":
            return
        cam = raybundle.camera_indices.reshape(-1)
        if cam.dtype != torch.int64:
            cam = cam.long()
        # SE3 (REF nerfacto_config/thermal_nerfacto.py:24 lists it): the same rotation, translation V(w) u — a dozen torch ops on
        # the [num_cameras, 6] table (autograd carries d(V u) back to u and w), then the SO3xR3 kernels per ray
        pose = _se3_translation(self.pose_adjustment) if self.config.mode == "SE3" else self.pose_adjustment
        raybundle.origins, raybundle.directions = _ApplyPoseAdjustment.apply(pose, cam, raybundle.origins, raybundle.directions)
