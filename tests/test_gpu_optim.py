"""GPU: HipAdam (csrc/tn_optim.hip through thermo_nerf_amd/optim.py) against torch.optim.Adam — the optimizer the reference's method
config names for every parameter group [REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:31-44] — and the deferred table update of
the training step (config.deferred_table_update: scatter + table Adam left on the side streams until the next field forward)."""
import copy

import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 63), (64,), (3, 5), (1,), (1000003,), (1 << 18, 2), (7, 6)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]


def test_hip_adam_follows_torch_adam_step_for_step():
    """Seven tensors (sizes that are not multiples of the float4 width, a 2 MB table, a 4 MB vector), three groups with their own lr /
    eps / weight decay (the reference's: 1e-2 / 1e-15 for fields and proposal networks, 6e-4 / 1e-8 / 1e-2 for camera_opt), a tensor
    that receives no gradient on some steps (the proposal networks: one step in six) — eight steps against torch.optim.Adam on the
    same gradients: parameters and both moments to fp32 rounding."""
    from thermo_nerf_amd.optim import HipAdam

    ours, ref = _params(), _params()

    def groups(ps):
        return [{"params": ps[:3]}, {"params": ps[3:6], "lr": 3e-3}, {"params": ps[6:], "lr": 6e-4, "eps": 1e-8, "weight_decay": 1e-2}]

    a = HipAdam(groups(ours), lr=1e-2, eps=1e-15)
    b = torch.optim.Adam(groups(ref), lr=1e-2, eps=1e-15)
    g = torch.Generator().manual_seed(7)
    for step in range(8):
        for k, (p, q) in enumerate(zip(ours, ref)):
            if k == 3 and step % 3:  # no gradient this step: skipped, its step counter stays behind
                p.grad = q.grad = None
                continue
            grad = (torch.randn(p.shape, generator=g) * (10.0 ** (step % 4 - 3))).to(DEV)
            if k == 1:
                grad[::2] = 0.0  # zero gradients keep moving through the moments
            p.grad, q.grad = grad.clone(), grad.clone()
        a.step()
        b.step()
    torch.cuda.synchronize()
    for k, (p, q) in enumerate(zip(ours, ref)):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), (k, (p - q).abs().max().item())
        sa, sb = a.state[p], b.state[q]
        assert float(sa["step"]) == float(sb["step"])
        # (torch's lerp contracts weight * (g - m) + m into one fma; this library is built with -ffp-contract=off: the moments
        # agree to an ulp of their LARGEST addend, not of a sum that cancelled)
        for name in ("exp_avg", "exp_avg_sq"):
            scale = float(sb[name].abs().max())
            assert torch.allclose(sa[name], sb[name], rtol=1e-5, atol=1e-6 * scale), (k, name, (sa[name] - sb[name]).abs().max().item())


def test_hip_adam_state_dicts_load_into_torch_adam_and_back():
    from thermo_nerf_amd.optim import HipAdam

    ours, ref = _params(1), _params(1)
    a, b = HipAdam(ours, lr=1e-2, eps=1e-15), torch.optim.Adam(ref, lr=1e-2, eps=1e-15)
    g = torch.Generator().manual_seed(3)

    def step_both():
        for p, q in zip(ours, ref):
            grad = torch.randn(p.shape, generator=g).to(DEV)
            p.grad, q.grad = grad.clone(), grad.clone()
        a.step()
        b.step()

    step_both()
    step_both()
    sd_a, sd_b = a.state_dict(), b.state_dict()
    assert set(sd_a["state"][0]) == set(sd_b["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert torch.is_tensor(sd_a["state"][0]["step"]) and float(sd_a["state"][0]["step"]) == 2.0
    a2, b2 = HipAdam(ours, lr=1e-2, eps=1e-15), torch.optim.Adam(ref, lr=1e-2, eps=1e-15)
    a2.load_state_dict(copy.deepcopy(sd_b))  # torch's state into ours ...
    b2.load_state_dict(copy.deepcopy(sd_a))  # ... and ours into torch's
    a, b = a2, b2
    step_both()
    torch.cuda.synchronize()
    for p, q in zip(ours, ref):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
        assert float(a.state[p]["step"]) == float(b.state[q]["step"]) == 3.0


def test_hip_adam_refuses_host_tensors_and_bad_list_sizes():
    from thermo_nerf_amd import _hip
    from thermo_nerf_amd.optim import HipAdam

    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipAdam([p]).step()
    lib = _hip.load()
    arr = (_hip.tn_adam_tensor * 1)()
    assert lib.tn_adam_step(arr, 0, None) == 0
    assert lib.tn_adam_step(arr, _hip.ADAM_MAX_TENSORS + 1, None) == -2  # TN_ERR_SHAPE
    arr[0].n = 8  # pointers NULL
    assert lib.tn_adam_step(arr, 1, None) == -1  # TN_ERR_NULL
    # more tensors than one launch's descriptor list holds: several launches, same result
    many = [torch.nn.Parameter(torch.full((5,), float(k), device=DEV)) for k in range(_hip.ADAM_MAX_TENSORS + 7)]
    for q in many:
        q.grad = torch.ones_like(q)
    HipAdam(many, lr=0.5).step()
    torch.cuda.synchronize()
    for k, q in enumerate(many):  # first Adam step: p -= lr * sign(g)
        assert torch.allclose(q, torch.full((5,), k - 0.5, device=DEV), atol=1e-6)


def _trainer(deferred: bool, impl: str = "hip", steps_seed: int = 0, **model_over):
    from thermo_nerf_amd import synthetic
    from thermo_nerf_amd.trainer import RayDataset, Trainer, TrainerConfig

    V, res = 6, 48
    cm, _, _ = helpers.build("init", 48, small=False, num_images=V, camera_optimizer_mode="SO3xR3", **model_over)
    cams = synthetic.orbit_cameras(res, res, list(range(V)), num_views=V, elevation_deg=[(-10.0, 20.0, 50.0)[v % 3] for v in range(V)])
    imgs, ths = [], []
    for i in range(V):
        rb = cams.generate_rays(i, device=DEV)
        im, th = synthetic.analytic_scene(rb.origins, rb.directions)
        imgs.append(im)
        ths.append(th)
    ds = RayDataset.from_images(cams, imgs, ths, DEV)
    model = copy.deepcopy(cm).to(DEV)
    tr = Trainer(model, ds, TrainerConfig(train_num_rays_per_batch=1024, optimizer_impl=impl, seed=steps_seed))
    if impl == "hip" and not deferred:
        model.config.deferred_table_update = False
        tr.optimizers["fields"]._deferred = set()
    return model, tr


def test_deferred_table_update_trains_like_the_joined_step():
    """config.deferred_table_update (the Trainer's default with HipAdam): the field's table scatter and Adam stay on the side streams
    and are joined by the NEXT field forward.  The same 14 Trainer iterations (the sampler's warm-up: update steps and frozen ones;
    same batches and jitter) with the update deferred, joined, and through torch.optim.Adam: per-step losses agree to the spread the
    scatter's atomics leave between two runs of one configuration, the table's first moment — linear in the gradients — to 1e-3, and
    reading the table right after train() is safe (train() joins)."""
    from thermo_nerf_amd import _hip

    runs = {}
    for tag, (deferred, impl) in {"deferred": (True, "hip"), "joined": (False, "hip"), "torch": (False, "torch")}.items():
        torch.manual_seed(11)
        model, tr = _trainer(deferred, impl)
        assert bool(getattr(model.config, "deferred_table_update", False)) == deferred
        losses = []
        for _ in range(14):
            loss, _, _ = tr.train_iteration(tr.step)
            tr.step += 1
            # host-side garbage between the steps: fresh allocations that would land in a block the side streams still use if
            # the deferred temporaries were released too early
            junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(4)]
            del junk
            losses.append(loss)
        if deferred:
            assert _hip.pending(torch.device(DEV)) is not None  # the last step's update is still out
        tr.train(0)  # the exit of train() joins
        assert _hip.pending(torch.device(DEV)) is None
        table = model.field.mlp_base.encoder.hash_table
        opt = tr.optimizers["fields"]
        runs[tag] = ([float(x) for x in losses], table.detach().clone(), opt.state[table]["exp_avg"].clone())
        assert torch.isfinite(runs[tag][1]).all()
    ref_losses, _, ref_m = runs["joined"]
    for tag in ("deferred", "torch"):
        losses, _, m = runs[tag]
        for k, (a, b) in enumerate(zip(losses, ref_losses)):
            assert abs(a - b) <= 2e-3 * abs(b), (tag, k, a, b)
        # (torch's fused Adam rounds otherwise than tn_adam_step; with eps = 1e-15 a last-bit difference in a near-zero moment is a
        # full +-lr step of either sign, which the next steps' gradients inherit: two IMPLEMENTATIONS drift apart faster than two
        # schedules of one)
        assert float((m - ref_m).norm() / ref_m.norm()) <= (2e-2 if tag == "deferred" else 0.15), tag
    assert ref_losses[-1] < 0.8 * ref_losses[0]


def test_deferred_update_is_joined_by_every_reader():
    """Whoever reads the table on the calling stream joins first: eval structs (get_outputs in eval mode, the engine), state_dict,
    Module.train(), the optimizer's own state_dict."""
    from thermo_nerf_amd import _hip
    from thermo_nerf_amd.rays import RayBundle

    dev = torch.device(DEV)
    model, tr = _trainer(True)
    o, d = helpers.rays(8, 8)

    def one_step():
        tr.train_iteration(tr.step)
        tr.step += 1
        assert _hip.pending(dev) is not None

    one_step()
    model.state_dict()
    assert _hip.pending(dev) is None
    one_step()
    tr.optimizers["fields"].state_dict()
    assert _hip.pending(dev) is None
    one_step()
    model.eval()
    assert _hip.pending(dev) is None
    with torch.no_grad():
        out = model(RayBundle(origins=o.to(DEV), directions=d.to(DEV)))
    assert torch.isfinite(out["rgb"]).all()
    model.train()
    one_step()
    path_losses = [float(tr.train_iteration(tr.step)[0])]  # the next training forward joins before its field launch
    torch.cuda.synchronize()
    assert path_losses[0] == path_losses[0]


def test_trainer_with_other_mlp_widths_follows_torch_adam():
    """thermo_nerf_amd.trainer.Trainer on a field with MLP widths other than 64 (field.staged: stage-by-stage forward and backward,
    layers above 64 wide included; the table scatter joined by the backward — the deferred update belongs to the fused step) with
    HipAdam against the same 30 iterations with torch.optim.Adam: the same losses while rounding has not separated the two runs,
    the same descent afterwards."""
    from thermo_nerf_amd import _hip

    runs = {}
    for impl in ("hip", "torch"):
        torch.manual_seed(3)
        model, tr = _trainer(impl == "hip", impl, hidden_dim=128, hidden_dim_color=96, hidden_dim_transient=32)
        assert model.field.staged
        losses = []
        for _ in range(30):
            loss, _, _ = tr.train_iteration(tr.step)
            tr.step += 1
            losses.append(float(loss))
        tr.train(0)
        assert _hip.pending(torch.device(DEV)) is None
        runs[impl] = losses
    a, b = runs["hip"], runs["torch"]
    for k in range(10):
        assert abs(a[k] - b[k]) <= 1e-3 * abs(b[k]), (k, a[k], b[k])
    assert abs(a[-1] - b[-1]) <= 0.05 * abs(b[-1]) and a[-1] < 0.3 * a[0], (a, b)
