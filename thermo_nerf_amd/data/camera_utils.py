"""The two nerfstudio ``camera_utils`` functions the Thermal dataparser calls [REF thermo_nerf/thermal_nerf/
thermal_dataparser.py:207-212] (nerfstudio 1.1.5; restated, the package is not importable here)."""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor


def rotation_matrix(a: Tensor, b: Tensor) -> Tensor:
    """NS camera_utils.rotation_matrix: the rotation taking direction ``a`` onto ``b`` (Rodrigues)."""
    a = a / torch.linalg.norm(a)
    b = b / torch.linalg.norm(b)
    v = torch.linalg.cross(a, b)
    eps = 1e-6
    if torch.sum(torch.abs(v)) < eps:  # (anti-)parallel: rotate about any axis orthogonal to a
        x = torch.tensor([1.0, 0.0, 0.0]) if abs(a[0]) < eps else torch.tensor([0.0, 1.0, 0.0])
        v = torch.linalg.cross(a, x)
    v = v / torch.linalg.norm(v)
    skew = torch.tensor([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    theta = torch.acos(torch.clip(torch.dot(a, b), -1, 1))
    return torch.eye(3) + torch.sin(theta) * skew + (1 - torch.cos(theta)) * (skew @ skew)


def auto_orient_and_center_poses(poses: Tensor, method: str = "up", center_method: str = "poses") -> Tuple[Tensor, Tensor]:
    """NS camera_utils.auto_orient_and_center_poses for the methods the reference can reach through
    NerfstudioDataParserConfig (orientation_method "up" (default) | "none", center_method "poses" (default) | "none").
    poses [N,3|4,4] -> (oriented poses [N,3,4], transform [3,4])."""
    poses = poses[:, :3, :4]
    origins = poses[..., :3, 3]
    mean_origin = torch.mean(origins, dim=0)
    if center_method == "poses":
        translation = mean_origin
    elif center_method == "none":
        translation = torch.zeros_like(mean_origin)
    else:
        raise NotImplementedError(f'center_method "{center_method}" (nerfstudio also has "focus")')
    if method == "up":
        up = torch.mean(poses[:, :3, 1], dim=0)
        up = up / torch.linalg.norm(up)
        rotation = rotation_matrix(up, torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([rotation, rotation @ -translation[..., None]], dim=-1)
    elif method == "none":
        transform = torch.eye(4)[:3]
        transform[:3, 3] = -translation
    else:
        raise NotImplementedError(f'orientation_method "{method}" (nerfstudio also has "pca" and "vertical")')
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).expand(poses.shape[0], 1, 4)
    oriented = transform @ torch.cat([poses, bottom], dim=1)
    return oriented, transform
