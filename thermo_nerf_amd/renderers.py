"""RGB / accumulation / depth renderers with nerfstudio's names (built at
[REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:186-191], called at :237-243,267-270), on MI355X."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from . import _hip
from .rays import RaySamples
from .thermal_nerf.thermal_renderer import composite_last_sample


class RGBRenderer(nn.Module):
    """NS RGBRenderer; only background_color="last_sample" (the NerfactoModelConfig default the reference keeps)."""

    def __init__(self, background_color="last_sample") -> None:
        super().__init__()
        if background_color != "last_sample":
            raise NotImplementedError('RGBRenderer kernels implement background_color="last_sample" '
                                      "(NerfactoModelConfig default used at REF thermal_nerf_model.py:187)")
        self.background_color = background_color

    def forward(self, rgb: Tensor, weights: Tensor, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None, background_color=None) -> Tensor:
        if ray_indices is not None and num_rays is not None:
            raise NotImplementedError("packed samples are not produced by ProposalNetworkSampler")
        return composite_last_sample(rgb, weights, training=self.training)

    def blend_background(self, image: Tensor) -> Tensor:
        """NS RGBRenderer.blend_background: an RGBA ground truth is composited over the background — black for the
        "last_sample" / "random" settings, as nerfstudio substitutes — and an RGB one passes through."""
        if image.shape[-1] < 4:
            return image
        return image[..., :3] * image[..., 3:]

    def blend_background_for_loss_computation(self, pred_image: Tensor, pred_accumulation: Tensor, gt_image: Tensor):
        """NS: with "last_sample" and an RGB (no alpha) ground truth there is nothing to blend."""
        return pred_image, gt_image[..., :3]


def _depth_call(weights: Tensor, ray_samples: RaySamples, want: str) -> Tensor:
    R, n = weights.shape[0], weights.shape[1]
    w = _hip.require_device_tensor(weights[..., 0], "weights")
    st = _hip.require_device_tensor(ray_samples.frustums.starts[..., 0], "starts")
    en = _hip.require_device_tensor(ray_samples.frustums.ends[..., 0], "ends")
    out = torch.empty((R,), dtype=torch.float32, device=w.device)
    scratch = torch.empty((2,), dtype=torch.float32, device=w.device) if want == "expected" else None
    lib = _hip.load()
    _hip.check(
        lib.tn_depth_fwd(w.data_ptr(), st.data_ptr(), en.data_ptr(), R, n,
                         out.data_ptr() if want == "accumulation" else None,
                         out.data_ptr() if want == "median" else None,
                         out.data_ptr() if want == "expected" else None,
                         _hip.ptr(scratch), _hip.current_stream()),
        "tn_depth_fwd",
    )
    return out[..., None]


class AccumulationRenderer(nn.Module):
    """NS AccumulationRenderer: sum of weights along the ray."""

    def forward(self, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        R, n = weights.shape[0], weights.shape[1]
        w = _hip.require_device_tensor(weights[..., 0], "weights")
        out = torch.empty((R,), dtype=torch.float32, device=w.device)
        lib = _hip.load()
        # starts/ends are only read for the depth outputs; pass the weights buffer as a placeholder
        _hip.check(lib.tn_depth_fwd(w.data_ptr(), w.data_ptr(), w.data_ptr(), R, n, out.data_ptr(), None, None, None,
                                    _hip.current_stream()), "tn_depth_fwd")
        return out[..., None]


class DepthRenderer(nn.Module):
    """NS DepthRenderer(method="median" | "expected")."""

    def __init__(self, method: str = "median") -> None:
        super().__init__()
        if method not in ("median", "expected"):
            raise ValueError(method)
        self.method = method

    def forward(self, weights: Tensor, ray_samples: RaySamples, ray_indices=None, num_rays=None) -> Tensor:
        if ray_indices is not None and num_rays is not None:
            raise NotImplementedError("Median depth calculation is not implemented for packed samples.")
        return _depth_call(weights, ray_samples, self.method)
