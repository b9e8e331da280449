"""The ``thermal-nerf`` method configuration [REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:17-48]: the values the reference
hands to nerfstudio's TrainerConfig / VanillaDataManagerConfig / ThermalNerfModelConfig / optimizers, on this package's own
trainer (thermo_nerf_amd.trainer).  tests/test_populate_modules.py holds it against what the reference's file sets when executed
(tests/golden/populate_modules.json, ``method_config``).  Not mirrored: the viewer entries (``viewer``, ``vis``: out of scope)."""
from __future__ import annotations

from dataclasses import dataclass, field

from ..trainer import OptimizerConfig, TrainerConfig
from .thermal_nerf_model import ThermalNerfModelConfig


@dataclass
class MethodConfig:
    method_name: str = "thermal-nerf"
    steps_per_eval_batch: int = 500  # REF :19
    eval_num_rays_per_batch: int = 4096  # REF :28
    trainer: TrainerConfig = field(default_factory=lambda: TrainerConfig(
        max_num_iterations=30000, steps_per_save=2000, train_num_rays_per_batch=4096, mixed_precision=True,  # REF :20-22, 27
        optimizers={
            "proposal_networks": OptimizerConfig(lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=200000),  # REF :33-38
            "fields": OptimizerConfig(lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=200000),  # REF :39-44
            # no "camera_opt" entry in the reference's file: see trainer.default_optimizers
            "camera_opt": OptimizerConfig(lr=6e-4, eps=1e-8, weight_decay=1e-2, lr_final=6e-6, max_steps=200000),
        }))
    model: ThermalNerfModelConfig = field(default_factory=lambda: ThermalNerfModelConfig(eval_num_rays_per_chunk=1 << 16))  # REF :30


thermal_nerf_config = MethodConfig()
