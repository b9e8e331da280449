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


def test_exp_map_se3_is_the_matrix_exponential_of_the_twist():
    """oracle.training.exp_map_SE3 (restating nerfstudio's lie_groups.exp_map_SE3) and the product's SE3 table transform against
    an independent computation: exp of the 4x4 twist matrix [[K(w), u], [0, 0]] = [[R(w), V(w) u], [0, 1]] (outside the
    theta^2 >= 1e-4 clamp, where the closed forms are exact)."""
    import torch

    from oracle import training as T
    from thermo_nerf_amd.camera_optimizer import _se3_exp, _so3xr3_exp

    g = torch.Generator().manual_seed(4)
    tangent = torch.randn(64, 6, generator=g, dtype=torch.float64) * torch.tensor([0.3, 0.3, 0.3, 1.2, 1.2, 1.2], dtype=torch.float64)
    tangent = tangent[(tangent[:, 3:] ** 2).sum(1) >= 1e-3]
    u, w = tangent[:, :3], tangent[:, 3:]
    twist = torch.zeros(tangent.shape[0], 4, 4, dtype=torch.float64)
    twist[:, 0, 1], twist[:, 0, 2] = -w[:, 2], w[:, 1]
    twist[:, 1, 0], twist[:, 1, 2] = w[:, 2], -w[:, 0]
    twist[:, 2, 0], twist[:, 2, 1] = -w[:, 1], w[:, 0]
    twist[:, :3, 3] = u
    want = torch.linalg.matrix_exp(twist)[:, :3, :4]
    assert (T.exp_map_SE3(tangent) - want).abs().max().item() <= 1e-12
    assert (_se3_exp(tangent) - want).abs().max().item() <= 1e-12
    # SO3xR3: the same rotation, the translation taken as is
    so3 = T.exp_map_SO3xR3(tangent)
    assert (so3[:, :, :3] - want[:, :, :3]).abs().max().item() <= 1e-12 and torch.equal(so3[:, :, 3], u)
    assert (_so3xr3_exp(tangent) - so3).abs().max().item() <= 1e-12


def test_lazy_ray_samples_materialise_to_the_eager_ones():
    """The training outputs' ray_samples_list entries (LazyRaySamples) are RaySamples whose [R,n,1] views, frustums and
    deltas appear on first use and equal what RayBundle.get_ray_samples builds eagerly [REF thermal_nerf_model.py:272-273]."""
    import torch

    from thermo_nerf_amd.rays import RayBundle, RaySamples
    from thermo_nerf_amd.samplers import LazyRaySamples, _samples_from_bins

    g = torch.Generator().manual_seed(3)
    R, n = 5, 7
    rb = RayBundle(origins=torch.rand(R, 3, generator=g), directions=torch.rand(R, 3, generator=g),
                   pixel_area=torch.rand(R, 1, generator=g), camera_indices=torch.arange(R)[:, None],
                   nears=torch.zeros(R, 1), fars=torch.ones(R, 1))
    spacing = torch.sort(torch.rand(R, n + 1, generator=g), dim=1).values
    eucl = spacing * 3.0
    lazy, eager = LazyRaySamples(rb, spacing, eucl, True), _samples_from_bins(rb, spacing, eucl, True)
    assert isinstance(lazy, RaySamples) and "deltas" not in lazy.__dict__ and lazy.uniform_spacing is True
    assert lazy.spacing_bins is spacing and lazy.eucl_bins is eucl
    assert lazy.shape == eager.shape == (R, n)
    for name in ("deltas", "spacing_starts", "spacing_ends", "camera_indices", "nears", "fars"):
        assert torch.equal(getattr(lazy, name), getattr(eager, name)), name
    for name in ("origins", "directions", "starts", "ends", "pixel_area"):
        assert torch.equal(getattr(lazy.frustums, name), getattr(eager.frustums, name)), name
    assert lazy.metadata is None and lazy.spacing_to_euclidean_fn is None
