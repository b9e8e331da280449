"""ThermalRenderer — alpha-composites per-sample temperatures into a thermal pixel on MI355X.

Interface mirror of [REF thermo_nerf/thermal_nerf/thermal_renderer.py:16-149]: same constructor, same
``forward(thermal, weights, ray_indices, num_rays, background_color)`` signature, same result
(sum_s w*T + T[last]*(1 - sum_s w); eval: nan_to_num first, clamp to [0,1] last), same error for packed
samples.  The compositing itself is ``tn_composite_fwd`` (one wave64 per ray).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from .. import _hip


def composite_last_sample(values: Tensor, weights: Tensor, training: bool) -> Tensor:
    """values [R,n,C], weights [R,n,1] -> [R,C] with the "last_sample" background."""
    if values.dim() != 3 or weights.dim() != 3:
        raise ValueError("expected non-packed samples: values [R,n,C], weights [R,n,1]")
    R, n, ch = values.shape
    v = _hip.require_device_tensor(values, "values")
    w = _hip.require_device_tensor(weights[..., 0], "weights")
    out = torch.empty((R, ch), dtype=torch.float32, device=v.device)
    lib = _hip.load()
    _hip.check(lib.tn_composite_fwd(v.data_ptr(), w.data_ptr(), R, n, ch, 1 if training else 0, out.data_ptr(),
                                    _hip.current_stream()), "tn_composite_fwd")
    return out


class ThermalRenderer(nn.Module):
    """Renders thermal images the way colour is rendered [REF thermal_renderer.py:16-24]."""

    def __init__(self, background_color="random") -> None:
        super().__init__()
        self.background_color = background_color

    @classmethod
    def combine_thermal(cls, thermal: Tensor, weights: Tensor, background_color="random",
                        ray_indices: Optional[Tensor] = None, num_rays: Optional[int] = None,
                        training: bool = True) -> Tensor:
        # REF :49 overrides whatever was passed: the background is always the last sample
        if ray_indices is not None and num_rays is not None:
            raise NotImplementedError("Background color 'last_sample' not implemented for packed samples.")
        return composite_last_sample(thermal, weights, training=training)

    def forward(self, thermal: Tensor, weights: Tensor, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None, background_color=None) -> Tensor:
        if background_color is None:
            background_color = self.background_color
        if ray_indices is not None and num_rays is not None:
            raise NotImplementedError("Background color 'last_sample' not implemented for packed samples.")
        # nan_to_num (REF :136-137) and clamp_ (REF :146-147) happen inside the kernel when not training
        return composite_last_sample(thermal, weights, training=self.training)
