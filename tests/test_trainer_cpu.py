"""Host logic of the training loop: nerfstudio's ExponentialDecay schedule and the reference's optimizer table
[REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:31-44]."""
import math

from thermo_nerf_amd.trainer import OptimizerConfig, TrainerConfig, default_optimizers, exponential_decay_multiplier


def test_reference_optimizer_table():
    opt = default_optimizers()
    for group in ("proposal_networks", "fields"):
        assert (opt[group].lr, opt[group].eps, opt[group].lr_final, opt[group].max_steps) == (1e-2, 1e-15, 1e-4, 200000)
    cfg = TrainerConfig()
    assert (cfg.max_num_iterations, cfg.steps_per_save, cfg.train_num_rays_per_batch) == (30000, 2000, 4096)


def test_exponential_decay_schedule():
    c = OptimizerConfig(lr=1e-2, lr_final=1e-4, max_steps=200000)
    assert math.isclose(exponential_decay_multiplier(c, 0), 1.0, rel_tol=1e-12)
    assert math.isclose(exponential_decay_multiplier(c, 200000), 1e-2, rel_tol=1e-12)
    assert math.isclose(exponential_decay_multiplier(c, 10**7), 1e-2, rel_tol=1e-12)  # clipped at max_steps
    assert math.isclose(exponential_decay_multiplier(c, 100000), 1e-1, rel_tol=1e-12)  # log-linear: geometric mean
    # the reference's 30 k-step run only reaches t = 0.15 of the 200 k-step schedule
    assert math.isclose(exponential_decay_multiplier(c, 30000) * 1e-2, 1e-2 * (1e-2) ** 0.15, rel_tol=1e-12)
    w = OptimizerConfig(lr=1e-2, lr_final=None, max_steps=1000, warmup_steps=100, lr_pre_warmup=1e-8)
    assert math.isclose(exponential_decay_multiplier(w, 0) * 1e-2, 1e-8, rel_tol=1e-9)
    assert math.isclose(exponential_decay_multiplier(w, 50) * 1e-2, 1e-8 + (1e-2 - 1e-8) * math.sin(math.pi / 4), rel_tol=1e-12)
    assert math.isclose(exponential_decay_multiplier(w, 100), 1.0, rel_tol=1e-12)
    assert math.isclose(exponential_decay_multiplier(w, 900), 1.0, rel_tol=1e-12)  # lr_final None: constant
