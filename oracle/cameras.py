"""TEST INFRASTRUCTURE — CPU restatement of nerfstudio 1.1.5 ``Cameras.generate_rays`` for a perspective camera
(parity unpinned: nerfstudio is not importable here; follows SURVEY §8f-1 and Cameras._generate_rays_from_coords)."""
from __future__ import annotations

import torch
from torch import Tensor

_EPS = torch.finfo(torch.float32).eps


def radial_and_tangential_undistort(coords: Tensor, distortion_params: Tensor, eps: float = 1e-3,
                                    max_iterations: int = 10) -> Tensor:
    """NS camera_utils.radial_and_tangential_undistort (+ _compute_residual_and_jacobian): Newton iterations on the
    OPENCV model, distortion_params = (k1, k2, k3, k4, p1, p2); coords [...,2] distorted -> undistorted."""
    k1, k2, k3, k4, p1, p2 = [distortion_params[..., i] for i in range(6)]
    xd, yd = coords[..., 0], coords[..., 1]
    x, y = xd, yd
    for _ in range(max_iterations):
        r = x * x + y * y
        d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
        fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
        fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
        d_r = k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4))
        d_x, d_y = 2.0 * x * d_r, 2.0 * y * d_r
        fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
        fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
        fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
        fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
        denominator = fy_x * fx_y - fx_x * fy_y
        x_numerator = fx * fy_y - fy * fx_y
        y_numerator = fy * fx_x - fx * fy_x
        step_x = torch.where(torch.abs(denominator) > eps, x_numerator / denominator, torch.zeros_like(denominator))
        step_y = torch.where(torch.abs(denominator) > eps, y_numerator / denominator, torch.zeros_like(denominator))
        x = x + step_x
        y = y + step_y
    return torch.stack([x, y], dim=-1)


def distort(coords: Tensor, distortion_params: Tensor) -> Tensor:
    """The forward OPENCV model the iteration above inverts (test helper)."""
    k1, k2, k3, k4, p1, p2 = [distortion_params[..., i] for i in range(6)]
    x, y = coords[..., 0], coords[..., 1]
    r = x * x + y * y
    d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
    return torch.stack([d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x), d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y)], dim=-1)


def generate_rays(c2w: Tensor, fx: float, fy: float, cx: float, cy: float, height: int, width: int,
                  distortion_params: Tensor = None):
    """c2w [3,4] -> origins [H,W,3], directions [H,W,3], pixel_area [H,W,1].  distortion_params [6] or None."""
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    y, x = ys + 0.5, xs + 0.5  # NS image_coords: pixel centres
    coord = torch.stack([(x - cx) / fx, -(y - cy) / fy], -1)
    coord_x = torch.stack([(x - cx + 1) / fx, -(y - cy) / fy], -1)
    coord_y = torch.stack([(x - cx) / fx, -(y - cy + 1) / fy], -1)
    stack = torch.stack([coord, coord_x, coord_y], dim=0)
    if distortion_params is not None and bool((distortion_params != 0).any()):  # NS: only cameras with non-zero parameters
        stack = radial_and_tangential_undistort(stack, distortion_params)
    dirs = torch.cat([stack, -torch.ones_like(stack[..., :1])], dim=-1)  # [3,H,W,3]
    rot = c2w[:3, :3]
    dirs = torch.sum(dirs[..., None, :] * rot, dim=-1)
    norm = torch.maximum(torch.linalg.vector_norm(dirs, dim=-1, keepdim=True), torch.tensor([_EPS]))
    dirs = dirs / norm
    origins = c2w[:3, 3].expand(height, width, 3)
    d = dirs[0]
    dx = torch.sqrt(torch.sum((d - dirs[1]) ** 2, dim=-1))
    dy = torch.sqrt(torch.sum((d - dirs[2]) ** 2, dim=-1))
    return origins, d, (dx * dy)[..., None]
