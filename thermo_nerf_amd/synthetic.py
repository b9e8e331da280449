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


def orbit_camera_rays(height: int, width: int, view: int = 0, num_views: int = 8, radius: float = 0.8,
                      fov_deg: float = 50.0, elevation_deg: float = 20.0) -> Tuple[Tensor, Tensor, Tensor]:
    """Pinhole camera on an orbit looking at the origin (nerfstudio/OpenGL convention: -z forward, +y up,
    pixel centres at +0.5).  Returns origins [H,W,3], unit directions [H,W,3], pixel_area [H,W,1] on the host."""
    az = 2.0 * math.pi * view / num_views
    el = math.radians(elevation_deg)
    eye = torch.tensor([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el)])
    fwd = -eye / eye.norm()
    up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    true_up = torch.linalg.cross(right, fwd)
    c2w = torch.stack([right, true_up, -fwd], dim=1)  # columns: x, y, z(back)
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


def model_state_dict_cpu(model: torch.nn.Module) -> Dict[str, Tensor]:
    """CPU fp32 copy of the state dict (nerfstudio key names) — what the oracle consumes in tests/bench."""
    return {k: v.detach().to("cpu").clone() for k, v in model.state_dict().items()}
