"""TEST INFRASTRUCTURE — CPU restatement of nerfstudio 1.1.5 ``Cameras.generate_rays`` for a perspective camera
(parity unpinned: nerfstudio is not importable here; follows SURVEY §8f-1 and Cameras._generate_rays_from_coords)."""
from __future__ import annotations

import torch
from torch import Tensor

_EPS = torch.finfo(torch.float32).eps


def generate_rays(c2w: Tensor, fx: float, fy: float, cx: float, cy: float, height: int, width: int):
    """c2w [3,4] -> origins [H,W,3], directions [H,W,3], pixel_area [H,W,1]."""
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    y, x = ys + 0.5, xs + 0.5  # NS image_coords: pixel centres
    coord = torch.stack([(x - cx) / fx, -(y - cy) / fy], -1)
    coord_x = torch.stack([(x - cx + 1) / fx, -(y - cy) / fy], -1)
    coord_y = torch.stack([(x - cx) / fx, -(y - cy + 1) / fy], -1)
    stack = torch.stack([coord, coord_x, coord_y], dim=0)
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
