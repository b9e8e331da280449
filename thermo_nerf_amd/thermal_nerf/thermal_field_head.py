"""Thermal field head [REF thermo_nerf/thermal_nerf/thermal_field_head.py:9-71,
thermo_nerf/thermal_nerf/thermal_field.py:18-30].

The head is a parameter holder: ``net`` is the ``nn.Linear(in_dim, 1)`` whose weights the fused heads kernel
(``tn_field_heads_fwd``) consumes; the state-dict key stays ``field_head_thermal.net.{weight,bias}``.
"""
from __future__ import annotations

from enum import Enum
from typing import Optional

from torch import nn


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
        raise RuntimeError(
            "ThermalFieldHead is evaluated inside tn_field_heads_fwd together with mlp_thermal; "
            "call ThermalNerfactoTField.get_outputs()."
        )


class ThermalFieldHead(BaseThermalFieldHead):
    """Thermal output head (out_dim 1, no activation) [REF thermal_field.py:18-30]."""

    def __init__(self, in_dim: Optional[int] = None) -> None:
        super().__init__(in_dim=in_dim, out_dim=1, field_head_name=FieldHeadNamesT.THERMAL, activation=None)
