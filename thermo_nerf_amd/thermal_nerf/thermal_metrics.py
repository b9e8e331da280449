"""Thermal mean absolute error in degrees [REF thermo_nerf/thermal_nerf/thermal_metrics.py:5-34].
Host-side bookkeeping (a handful of elementwise torch ops on whatever device the images live on)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor


def mae_thermal(gt: Tensor, pred: Tensor, cold_flag: bool, max_temperature: float, min_temperature: float,
                threshold: Optional[float] = None) -> Tensor:
    """De-normalise ``gt``/``pred`` from [0,1] to [min_temperature, max_temperature] and return the mean |error|.
    With ``threshold`` only the foreground (gt above it, or below it when ``cold_flag``) is scored."""
    if threshold:
        keep = torch.where(gt < threshold) if cold_flag else torch.where(gt > threshold)
        gt, pred = gt[keep], pred[keep]
    span = max_temperature - min_temperature
    gt_deg = gt * span + min_temperature
    pred_deg = pred * span + min_temperature
    return torch.mean(torch.abs(gt_deg - pred_deg))
