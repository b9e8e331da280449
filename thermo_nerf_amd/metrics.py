"""Image metrics of the eval harness on the device: what ``get_image_metrics_and_images`` asks of
torchmetrics [REF thermal_nerf_model.py:195-200,362-363]."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _hip


def psnr(pred: Tensor, gt: Tensor) -> Tensor:
    """torchmetrics PeakSignalNoiseRatio(data_range=1.0) [REF thermal_nerf_model.py:200]: 10 log10(1 / MSE)."""
    return 10.0 * torch.log10(1.0 / torch.mean((pred - gt) ** 2))


def ssim(pred: Tensor, gt: Tensor) -> Tensor:
    """torchmetrics ``structural_similarity_index_measure`` with its defaults (``tn_ssim_fwd``) on one frame in the
    renderers' layout: pred / gt [H,W,C] device tensors.  The reference calls it on [1,C,H,W] views of the same data
    [REF thermal_nerf_model.py:355-363]; the window must fit: H, W >= 11."""
    if pred.shape != gt.shape or pred.dim() != 3:
        raise ValueError(f"ssim expects two [H,W,C] images of one shape, got {tuple(pred.shape)} and {tuple(gt.shape)}")
    p = _hip.require_device_tensor(pred.contiguous(), "pred")
    g = _hip.require_device_tensor(gt.to(p).contiguous(), "gt")
    h, w, c = p.shape
    lib = _hip.load()
    (pmin, pmax), (gmin, gmax) = torch.aminmax(p), torch.aminmax(g)
    data_range = float(torch.maximum(pmax - pmin, gmax - gmin))  # torchmetrics' data_range=None
    need = lib.tn_ssim_workspace_bytes(h, w, c)
    if need == 0:
        raise ValueError(f"ssim needs images of at least 11 x 11 pixels, got {h} x {w}")
    ws = torch.empty(need, dtype=torch.uint8, device=p.device)
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    _hip.check(lib.tn_ssim_fwd(p.data_ptr(), g.data_ptr(), h, w, c, data_range, ws.data_ptr(), need, out.data_ptr(),
                               _hip.current_stream()), "tn_ssim_fwd")
    return out[0]
