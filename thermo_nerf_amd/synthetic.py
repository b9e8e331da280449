"""Synthetic scenes for parity tests and the benchmark: counter-hash weights that both boxes
rebuild bit-identically, and pinhole orbit cameras.

Everything here is generated on the HOST with integer hashing + exactly-rounded float ops, then moved to the
device, so the build container, the GPU box and the CPU oracle see the same bits without shipping 67 MB tables.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
from torch import Tensor

SEED = 0x7E3A0


def counter_uniform(numel: int, stream: int, seed: int = SEED) -> Tensor:
    """U[0,1) float32 from a 32-bit integer hash of (index, stream, seed); pure integer ops + one exact scale."""
    idx = torch.arange(numel, dtype=torch.int64)
    x = (idx + (stream * 0x9E3779B1 + seed * 0x85EBCA77)) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return (x >> 8).to(torch.float32) * (1.0 / 16777216.0)


def counter_normalish(numel: int, stream: int, seed: int = SEED) -> Tensor:
    """Approximately N(0,1): Irwin-Hall sum of 12 uniforms minus 6 (fixed add order => reproducible)."""
    acc = torch.zeros(numel, dtype=torch.float32)
    for k in range(12):
        acc = acc + counter_uniform(numel, stream * 16 + k + 1000, seed)
    return acc - 6.0


def fill_model_(model: torch.nn.Module, kind: str = "init", seed: int = SEED) -> torch.nn.Module:
    """Overwrite every parameter of a ThermalNerfModel with counter-hash values (call on the CPU model).

    kind="init":   nerfstudio's init distributions — hash tables U(-1,1)*1e-3, Linear U(+-1/sqrt(fan_in)),
                   appearance embedding ~N(0,1)  (SURVEY A.4/A.5).
    kind="stress": hash tables U(-1,1) and the density output bias +2, so sigma spans e^(+-3): saturated
                   transmittance, sum(w) ~ 1, sharp PDFs.
    kind="scene":  "stress" tables scaled 0.3 + density bias +1: a mid-regime mix of empty and opaque rays.
    """
    if kind not in ("init", "stress", "scene"):
        raise ValueError(kind)
    table_scale = {"init": 1e-3, "stress": 1.0, "scene": 0.3}[kind]
    with torch.no_grad():
        for stream, (name, p) in enumerate(sorted(model.named_parameters(), key=lambda kv: kv[0])):
            if p.numel() == 0:
                continue
            n = p.numel()
            if name.endswith("hash_table"):
                v = (counter_uniform(n, stream, seed) * 2 - 1) * table_scale
            elif name.endswith("embedding.weight"):
                v = counter_normalish(n, stream, seed)
            elif name.endswith("pose_adjustment"):
                v = torch.zeros(n)
            else:
                fan_in = p.shape[1] if p.dim() == 2 else None
                if fan_in is None:  # bias: fan_in of the matching weight
                    w = dict(model.named_parameters())[name[: -len("bias")] + "weight"]
                    fan_in = w.shape[1]
                v = (counter_uniform(n, stream, seed) * 2 - 1) * (1.0 / math.sqrt(fan_in))
            p.copy_(v.view_as(p))
        if kind in ("stress", "scene"):
            bump = 2.0 if kind == "stress" else 1.0
            model.field.mlp_base.mlp.layers[1].bias[0] += bump
            for net in model.proposal_networks:
                net.mlp_base.mlp.layers[1].bias[0] += bump
    return model


def orbit_pose(view: int, num_views: int = 8, radius: float = 0.8, elevation_deg: float = 20.0) -> Tuple[Tensor, Tensor]:
    """(rotation [3,3] with columns x, y, z(back); eye [3]) of a camera on an orbit looking at the origin
    (nerfstudio/OpenGL convention: -z forward, +y up)."""
    az = 2.0 * math.pi * view / num_views
    el = math.radians(elevation_deg)
    eye = torch.tensor([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el)])
    fwd = -eye / eye.norm()
    up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    true_up = torch.linalg.cross(right, fwd)
    return torch.stack([right, true_up, -fwd], dim=1), eye


def orbit_cameras(height: int, width: int, views, num_views: int = 8, radius: float = 0.8, fov_deg: float = 50.0,
                  elevation_deg=20.0):
    """The same orbit as ``orbit_camera_rays`` as a ``Cameras`` object: one camera per entry of ``views`` (azimuth index,
    may be fractional); ``elevation_deg`` is a number or one value per view."""
    from .cameras import Cameras

    c2w = []
    for k, v in enumerate(views):
        el = elevation_deg[k] if isinstance(elevation_deg, (list, tuple)) else elevation_deg
        rot, eye = orbit_pose(v, num_views, radius, el)
        c2w.append(torch.cat([rot, eye[:, None]], dim=1))
    f = 0.5 * width / math.tan(0.5 * math.radians(fov_deg))
    n = len(c2w)
    return Cameras(camera_to_worlds=torch.stack(c2w).float(), fx=torch.full((n,), f), fy=torch.full((n,), f),
                   cx=width / 2.0, cy=height / 2.0, height=height, width=width)


def orbit_camera_rays(height: int, width: int, view: int = 0, num_views: int = 8, radius: float = 0.8,
                      fov_deg: float = 50.0, elevation_deg: float = 20.0) -> Tuple[Tensor, Tensor, Tensor]:
    """Pinhole camera on an orbit looking at the origin (nerfstudio/OpenGL convention: -z forward, +y up,
    pixel centres at +0.5).  Returns origins [H,W,3], unit directions [H,W,3], pixel_area [H,W,1] on the host."""
    c2w, eye = orbit_pose(view, num_views, radius, elevation_deg)
    fx = fy = 0.5 * width / math.tan(0.5 * math.radians(fov_deg))
    cx, cy = width / 2.0, height / 2.0
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32) + 0.5,
                            torch.arange(width, dtype=torch.float32) + 0.5, indexing="ij")
    cam = torch.stack([(xs - cx) / fx, -(ys - cy) / fy, -torch.ones_like(xs)], dim=-1)
    d = cam @ c2w.T
    norm = d.norm(dim=-1, keepdim=True)
    d = d / norm
    o = eye.expand(height, width, 3).contiguous()
    pixel_area = (1.0 / (fx * fy)) * torch.ones(height, width, 1)
    return o.float(), d.float().contiguous(), pixel_area.float()


def random_pixel_rays(rays: int, height: int = 800, width: int = 800, num_views: int = 8, seed: int = 0, radius: float = 0.8,
                      fov_deg: float = 50.0, elevation_deg: float = 20.0) -> Tuple[Tensor, Tensor, Tensor]:
    """A training batch the way nerfstudio's PixelSampler draws it: ``rays`` pixels uniformly at random over ALL
    ``num_views`` images of the orbit (not a patch of one image), same pinhole model as ``orbit_camera_rays``.
    Returns origins [rays,3], unit directions [rays,3], camera indices [rays,1] (int64) on the host."""
    g = torch.Generator().manual_seed(seed)
    view = torch.randint(0, num_views, (rays,), generator=g)
    ys = torch.randint(0, height, (rays,), generator=g).float() + 0.5
    xs = torch.randint(0, width, (rays,), generator=g).float() + 0.5
    f = 0.5 * width / math.tan(0.5 * math.radians(fov_deg))
    cam = torch.stack([(xs - width / 2.0) / f, -(ys - height / 2.0) / f, -torch.ones_like(xs)], dim=-1)
    o = torch.empty(rays, 3)
    d = torch.empty(rays, 3)
    for v in range(num_views):
        m = view == v
        c2w, eye = orbit_pose(v, num_views, radius, elevation_deg)
        dv = cam[m] @ c2w.T
        d[m] = dv / dv.norm(dim=-1, keepdim=True)
        o[m] = eye
    return o.float().contiguous(), d.float().contiguous(), view[:, None].contiguous()


def model_state_dict_cpu(model: torch.nn.Module) -> Dict[str, Tensor]:
    """CPU fp32 copy of the state dict (nerfstudio key names) — what the oracle consumes in tests/bench."""
    return {k: v.detach().to("cpu").clone() for k, v in model.state_dict().items()}


def analytic_scene(origins: Tensor, directions: Tensor) -> Tuple[Tensor, Tensor]:
    """Ground truth of a closed-form RGB + thermal scene for training tests / demos (no dataset on this box):
    a textured sphere (radius 0.3, warm top / cold bottom) in front of a direction-dependent backdrop.
    origins / unit directions [...,3] -> (rgb [...,3], thermal [...,1]) in [0,1]."""
    o, d = origins, directions
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - 0.3 * 0.3
    disc = b * b - c
    hit = (disc > 0) & (-b - torch.sqrt(disc.clamp_min(0)) > 0)
    t = (-b - torch.sqrt(disc.clamp_min(0)))
    p = o + d * t[..., None]
    n = p / 0.3
    shade = 0.35 + 0.65 * (n * torch.tensor([0.3, 0.5, 0.8]).to(n)).sum(-1).clamp(0, 1)
    stripes = 0.5 + 0.5 * torch.sin(18.0 * p[..., 2:3] + 6.0 * torch.atan2(p[..., 1:2], p[..., 0:1]))
    albedo = torch.cat([0.9 - 0.5 * stripes, 0.3 + 0.5 * stripes, 0.25 + 0.2 * n[..., 0:1].abs()], dim=-1)
    rgb_s = (albedo * shade[..., None]).clamp(0, 1)
    th_s = (0.55 + 0.4 * n[..., 2:3]).clamp(0, 1)
    rgb_b = torch.stack([0.25 + 0.2 * d[..., 2], 0.3 + 0.15 * d[..., 0], 0.45 + 0.25 * d[..., 1]], dim=-1).clamp(0, 1)
    th_b = torch.full_like(th_s, 0.15)
    h = hit[..., None]
    return torch.where(h, rgb_s, rgb_b), torch.where(h, th_s, th_b)


def analytic_room_scene(origins: Tensor, directions: Tensor, room_radius: float = 2.0) -> Tuple[Tensor, Tensor]:
    """The sphere of ``analytic_scene`` inside a textured spherical room of radius ``room_radius`` (cameras inside it).  The
    backdrop of ``analytic_scene`` is a function of the ray DIRECTION only — geometry at infinity, which a radiance field with a
    view-dependent colour head can paint at any depth (a long optimisation at full capacity then parks it right in front of
    the cameras and held-out views fall apart); here every pixel is a point in WORLD space, so parallax between the training
    views pins the geometry, as it does for a captured scene.  Same contract as ``analytic_scene``."""
    o, d = origins, directions
    rgb_s, th_s = analytic_scene(o, d)
    b = (o * d).sum(-1)
    hit = ((b * b - ((o * o).sum(-1) - 0.3 * 0.3)) > 0) & (-b - torch.sqrt((b * b - ((o * o).sum(-1) - 0.09)).clamp_min(0)) > 0)
    t_wall = -b + torch.sqrt((b * b - ((o * o).sum(-1) - room_radius * room_radius)).clamp_min(0))
    p = o + d * t_wall[..., None]
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    tiles = torch.sin(3.0 * x + 1.0) * torch.sin(3.0 * y - 0.5) * torch.sin(3.0 * z + 2.0)
    rgb_w = torch.stack([0.45 + 0.30 * torch.sin(2.2 * x + 0.7 * z) + 0.15 * tiles,
                         0.40 + 0.25 * torch.sin(2.6 * y - 0.9 * x) - 0.15 * tiles,
                         0.50 + 0.30 * torch.sin(2.4 * z + 1.1 * y)], dim=-1).clamp(0, 1)
    th_w = (0.22 + 0.10 * torch.sin(1.7 * z + 0.6 * x) + 0.05 * tiles)[..., None].clamp(0, 1)
    h = hit[..., None]
    return torch.where(h, rgb_s, rgb_w), torch.where(h, th_s, th_w)


def spiral_cameras(height: int, width: int, count: int, radius: float = 0.8, fov_deg: float = 50.0,
                   elevation_range=(-15.0, 65.0), phase: float = 0.0):
    """``count`` cameras looking at the origin from a golden-angle spiral over a band of elevations (even coverage of the band;
    ``phase`` shifts the spiral so that a second set falls between the first one's cameras)."""
    lo, hi = (math.sin(math.radians(e)) for e in elevation_range)
    views, elev = [], []
    for k in range(count):
        u = (k + 0.5 + 0.37 * phase) / count
        elev.append(math.degrees(math.asin(lo + (hi - lo) * u)))
        views.append(((k + phase) * 137.50776405) % 360.0)
    return orbit_cameras(height, width, views, num_views=360, radius=radius, fov_deg=fov_deg, elevation_deg=elev)
