"""TEST INFRASTRUCTURE (CPU oracle): restatement of the torchmetrics calls the reference's models make on a rendered frame.

``ssim`` restates ``torchmetrics.functional.structural_similarity_index_measure`` (torchmetrics 1.7.2, REF uv.lock:5468-5469;
called with its defaults by NS NerfactoModel.get_image_metrics_and_images and REF thermal_nerf_model.py:363): ``_ssim_update``
with gaussian_kernel=True, sigma=1.5, kernel_size=11, data_range=None, k1=0.01, k2=0.03.  torchmetrics is not importable
here: parity unpinned (the restatement follows the published algorithm; tests check it against an independent float64
computation with scipy.ndimage)."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def _gaussian(kernel_size: int, sigma: float, dtype) -> Tensor:
    dist = torch.arange(start=(1 - kernel_size) / 2, end=(1 + kernel_size) / 2, step=1, dtype=dtype)
    gauss = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return (gauss / gauss.sum()).unsqueeze(dim=0)  # (1, kernel_size)


def ssim(preds: Tensor, target: Tensor, sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03) -> Tensor:
    """preds / target [B,C,H,W] (what the models pass after moveaxis).  Returns the mean over the batch (reduction
    "elementwise_mean")."""
    data_range = torch.max(preds.max() - preds.min(), target.max() - target.min())
    c1 = torch.pow(k1 * data_range, 2)
    c2 = torch.pow(k2 * data_range, 2)
    channel = preds.size(1)
    ks = int(3.5 * sigma + 0.5) * 2 + 1  # 11
    pad = (ks - 1) // 2
    preds = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    target = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    g = _gaussian(ks, sigma, preds.dtype)
    kernel = torch.matmul(g.t(), g).expand(channel, 1, ks, ks)
    input_list = torch.cat((preds, target, preds * preds, target * target, preds * target))
    outputs = F.conv2d(input_list, kernel, groups=channel)
    o = outputs.split(preds.shape[0])
    mu_pred_sq, mu_target_sq, mu_pred_target = o[0].pow(2), o[1].pow(2), o[0] * o[1]
    sigma_pred_sq = torch.clamp(o[2] - mu_pred_sq, min=0.0)
    sigma_target_sq = torch.clamp(o[3] - mu_target_sq, min=0.0)
    sigma_pred_target = o[4] - mu_pred_target
    upper = 2 * sigma_pred_target + c2
    lower = sigma_pred_sq + sigma_target_sq + c2
    full = ((2 * mu_pred_target + c1) * upper) / ((mu_pred_sq + mu_target_sq + c1) * lower)
    idx = full[..., pad:-pad, pad:-pad]
    return idx.reshape(idx.shape[0], -1).mean(-1).mean()


def psnr(preds: Tensor, target: Tensor, data_range: float = 1.0) -> Tensor:
    """torchmetrics PeakSignalNoiseRatio(data_range=1.0) [REF thermal_nerf_model.py:200]."""
    return 10.0 * torch.log10(data_range ** 2 / torch.mean((preds - target) ** 2))


def thermal_image_metrics(gt_thermal: Tensor, pred_thermal: Tensor, cold: bool, max_temperature: float, min_temperature: float,
                          threshold=None) -> dict:
    """What ThermalNerfModel.get_image_metrics_and_images adds for the thermal modality [REF thermal_nerf_model.py:354-391]:
    [H,W,1] images moved to [1,1,H,W], PSNR / SSIM on them, the two MAE figures in degrees (threshold only on the foreground
    one).  Key names and values pinned by the reference's own method (fixture G7, ``metrics.*``); LPIPS (a pretrained network)
    is outside the oracle."""
    from .hotpath import mae_thermal

    gt = torch.moveaxis(gt_thermal, -1, 0)[None, ...]
    pr = torch.moveaxis(pred_thermal, -1, 0)[None, ...]
    return {
        "psnr_thermal": float(psnr(gt, pr)),
        "ssim_thermal": float(ssim(gt, pr)),
        "mae_thermal_foreground": float(mae_thermal(gt, pr, cold, max_temperature, min_temperature, threshold)),
        "mae_thermal": float(mae_thermal(gt, pr, cold, max_temperature, min_temperature, None)),
    }
