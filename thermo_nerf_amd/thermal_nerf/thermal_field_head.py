"""Thermal field head [REF thermo_nerf/thermal_nerf/thermal_field_head.py:9-71,
thermo_nerf/thermal_nerf/thermal_field.py:18-30].

``net`` is the ``nn.Linear(in_dim, 1)`` whose weights the fused heads kernel (``tn_field_heads_fwd``) consumes on the
field's own path; called on its own, the head runs ``tn_linear_fwd`` (no activation) like the reference's
``self.net(in_tensor)``.  The state-dict key stays ``field_head_thermal.net.{weight,bias}``.
"""
from __future__ import annotations

from enum import Enum
from typing import Optional

import torch
from torch import nn

from .. import _hip


class FieldHeadNamesT(Enum):
    """Thermal field outputs [REF thermal_field_head.py:9-12]."""

    THERMAL = "thermal"


class BaseThermalFieldHead(nn.Module):
    """Linear head, optional activation (None for thermal) [REF thermal_field_head.py:15-71]."""

    def __init__(self, out_dim: int, field_head_name: FieldHeadNamesT, in_dim: Optional[int] = None,
                 activation=None) -> None:
        super().__init__()
        if activation is not None:
            raise NotImplementedError("the thermal head is built with activation=None (REF thermal_field.py:25-30)")
        self.out_dim = out_dim
        self.activation = activation
        self.field_head_name = field_head_name
        self.net: Optional[nn.Linear] = None
        if in_dim is not None:
            self.set_in_dim(in_dim)

    def set_in_dim(self, in_dim: int) -> None:
        self.in_dim = in_dim
        self.net = nn.Linear(self.in_dim, self.out_dim)

    def forward(self, in_tensor):
        if self.net is None:
            raise SystemError("in_dim not set. Must be provided to constructor, or set_in_dim() should be called.")
        # REF thermal_field_head.py:66-69: out = self.net(in_tensor); activation is None for the thermal head
        x = _hip.require_device_tensor(in_tensor.reshape(-1, self.in_dim), "in_tensor")
        n = x.shape[0]
        y = torch.empty((n, self.out_dim), dtype=torch.float32, device=x.device)
        _hip.check(_hip.load().tn_linear_fwd(x.data_ptr(), self.in_dim, _hip.make_linear(self.net), 0, n, y.data_ptr(),
                                             self.out_dim, _hip.current_stream()), "tn_linear_fwd")
        return y.view(*in_tensor.shape[:-1], self.out_dim)


class ThermalFieldHead(BaseThermalFieldHead):
    """Thermal output head (out_dim 1, no activation) [REF thermal_field.py:18-30]."""

    def __init__(self, in_dim: Optional[int] = None) -> None:
        super().__init__(in_dim=in_dim, out_dim=1, field_head_name=FieldHeadNamesT.THERMAL, activation=None)
