"""Field components with nerfstudio's module/parameter names (so a nerfstudio checkpoint's state-dict keys
load unchanged — SURVEY §8b "State-dict names"), evaluated by the HIP kernels.

  HashEncoding, MLP, MLPWithHashEncoding   NS field_components (torch path = what the locked env runs)
  Embedding                                NS field_components.embedding.Embedding
  HashMLPDensityField                      NS fields.density_fields, built at [REF thermal_nerf_model.py:136-149]
  FieldHeadNames                           NS field_components.field_heads.FieldHeadNames
"""
from __future__ import annotations

from enum import Enum
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from . import _hip
from .rays import RaySamples
from .scene import SceneContraction


class FieldHeadNames(Enum):
    """NS FieldHeadNames (the members the reference path reads [REF thermal_field.py:168-169,197])."""

    RGB = "rgb"
    DENSITY = "density"
    NORMALS = "normals"
    PRED_NORMALS = "pred_normals"
    TRANSIENT_RGB = "transient_rgb"
    TRANSIENT_DENSITY = "transient_density"


class HashEncoding(nn.Module):
    """NS HashEncoding, torch path: ``hash_table`` [L*T, F] parameter + ``scalings`` [L] buffer."""

    def __init__(self, num_levels: int = 16, min_res: int = 16, max_res: int = 1024, log2_hashmap_size: int = 19,
                 features_per_level: int = 2, hash_init_scale: float = 0.001) -> None:
        super().__init__()
        if features_per_level != 2:
            raise NotImplementedError("the HIP hash-grid kernels implement features_per_level == 2")
        if not 1 <= num_levels <= _hip.TN_MAX_LEVELS:
            raise ValueError(f"num_levels must be in [1, {_hip.TN_MAX_LEVELS}]")
        self.num_levels = num_levels
        self.min_res = min_res
        self.max_res = max_res
        self.features_per_level = features_per_level
        self.log2_hashmap_size = log2_hashmap_size
        self.hash_table_size = 2**log2_hashmap_size
        levels = torch.arange(num_levels)
        self.growth_factor = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
        # evaluated by torch in float32, exactly as nerfstudio does; the kernels take these values verbatim
        self.register_buffer("scalings", torch.floor(min_res * self.growth_factor**levels))
        table = torch.rand(size=(self.hash_table_size * num_levels, features_per_level)) * 2 - 1
        self.hash_table = nn.Parameter(table * hash_init_scale)
        self._dense: Optional[Tensor] = None
        self._dense_grid: Optional[_hip.tn_hashgrid] = None
        self._dense_key = None

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def c_struct(self, dense_budget_bytes: int = 0) -> _hip.tn_hashgrid:
        table = _hip.require_device_tensor(self.hash_table.detach(), "hash_table")
        g = _hip.tn_hashgrid()
        g.table = table.data_ptr()
        sc = _hip.host_values(self, "scalings")  # cached on the module: no device read-back per call
        for i in range(self.num_levels):
            g.scalings[i] = sc[i]
        g.num_levels = self.num_levels
        g.log2_hashmap_size = self.log2_hashmap_size
        g.dense = None
        g.num_dense_levels = 0
        if dense_budget_bytes > 0:
            key = (table.data_ptr(), table._version, dense_budget_bytes)
            if self._dense_key != key:
                lib = _hip.load()
                nbytes = lib.tn_hashgrid_prepare_bytes(g, dense_budget_bytes)
                out = _hip.tn_hashgrid()
                self._dense = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=table.device)
                _hip.check(lib.tn_hashgrid_prepare(g, out, self._dense.data_ptr(), nbytes, _hip.current_stream()),
                           "tn_hashgrid_prepare")
                self._dense_grid = out
                self._dense_key = key
            return self._dense_grid
        return g


class MLP(nn.Module):
    """NS MLP, torch path: ``layers`` = ModuleList of nn.Linear WITH bias."""

    def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: Optional[int] = None) -> None:
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim if out_dim is not None else layer_width
        dims = [in_dim] + [layer_width] * (num_layers - 1) + [self.out_dim]
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers)])

    def get_out_dim(self) -> int:
        return self.out_dim


class MLPWithHashEncoding(nn.Module):
    """NS MLPWithHashEncoding, torch path: ``encoder`` (HashEncoding) + ``mlp`` (MLP)."""

    def __init__(self, num_levels: int, min_res: int, max_res: int, log2_hashmap_size: int, features_per_level: int,
                 num_layers: int, layer_width: int, out_dim: int) -> None:
        super().__init__()
        self.encoder = HashEncoding(num_levels, min_res, max_res, log2_hashmap_size, features_per_level)
        self.mlp = MLP(self.encoder.get_out_dim(), num_layers, layer_width, out_dim)


class Embedding(nn.Module):
    """NS Embedding: wraps nn.Embedding; ``mean(dim)`` is the mean of the weight."""

    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.embedding = nn.Embedding(in_dim, out_dim)

    def mean(self, dim=0) -> Tensor:
        return self.embedding.weight.mean(dim)


class HashMLPDensityField(nn.Module):
    """NS HashMLPDensityField (use_linear=False): the proposal networks [REF thermal_nerf_model.py:140-149]."""

    def __init__(self, aabb: Tensor, num_layers: int = 2, hidden_dim: int = 64,
                 spatial_distortion: Optional[SceneContraction] = None, use_linear: bool = False, num_levels: int = 8,
                 max_res: int = 1024, base_res: int = 16, log2_hashmap_size: int = 18, features_per_level: int = 2,
                 average_init_density: float = 1.0, implementation: str = "hip") -> None:
        super().__init__()
        if use_linear:
            raise NotImplementedError("use_linear=True is not on the ThermoNeRF path (proposal_net_args_list sets False)")
        if num_layers != 2:
            raise NotImplementedError("HashMLPDensityField kernels implement num_layers == 2")
        self.register_buffer("aabb", aabb.clone().float())
        self.spatial_distortion = spatial_distortion
        self.use_linear = use_linear
        self.average_init_density = average_init_density
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.mlp_base = MLPWithHashEncoding(num_levels, base_res, max_res, log2_hashmap_size, features_per_level,
                                            num_layers, hidden_dim, 1)
        self.dense_budget_bytes = 0

    def c_struct(self, dense: bool = True) -> _hip.tn_density_field:
        """``dense=False`` (the training path): never the dense re-layout of the coarse levels — it is a derived copy of
        the table, and the table is about to change."""
        f = _hip.tn_density_field()
        f.grid = self.mlp_base.encoder.c_struct(self.dense_budget_bytes if dense else 0)
        f.l0 = _hip.make_linear(self.mlp_base.mlp.layers[0])
        f.l1 = _hip.make_linear(self.mlp_base.mlp.layers[1])
        f.space = _hip.make_space(self.spatial_distortion is not None, self.aabb, owner=self)
        f.average_init_density = float(self.average_init_density)
        return f

    def train_struct(self) -> _hip.tn_density_field:
        """``c_struct(dense=False)``, kept for as long as the parameters keep their storage (an optimizer updates them in place,
        so the pointers in the struct stay right): a training step asks for it several times.  Read-only for the caller."""
        plist = self.__dict__.get("_tn_plist")
        if plist is None:
            plist = self.__dict__["_tn_plist"] = list(self.parameters())
        # (the struct holds the scene box by VALUE: an in-place change of the buffer must rebuild it)
        ptrs = tuple([p.data_ptr() for p in plist]) + (self.aabb.data_ptr(), self.aabb._version, self.spatial_distortion is not None,
                                                        float(self.average_init_density))
        hit = self.__dict__.get("_tn_train_struct")
        if hit is None or hit[0] != ptrs:
            hit = self.__dict__["_tn_train_struct"] = (ptrs, self.c_struct(dense=False))
        return hit[1]

    def density_fn(self, positions: Tensor) -> Tensor:
        """NS Field.density_fn: positions [...,3] -> density [...,1]."""
        pos = _hip.require_device_tensor(positions, "positions")
        flat = pos.reshape(-1, 3)
        out = torch.empty((flat.shape[0],), dtype=torch.float32, device=flat.device)
        lib = _hip.load()
        _hip.check(lib.tn_density_fwd(self.c_struct(), flat.data_ptr(), flat.shape[0], out.data_ptr(),
                                      _hip.current_stream()), "tn_density_fwd")
        return out.view(*positions.shape[:-1], 1)

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, None]:
        return self.density_fn(ray_samples.frustums.get_positions()), None

    def forward(self, ray_samples: RaySamples):
        density, _ = self.get_density(ray_samples)
        return {FieldHeadNames.DENSITY: density}
