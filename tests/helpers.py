"""Shared test plumbing: build a model + the matching oracle config / CPU state dict."""
from __future__ import annotations

import functools
from typing import Dict, Tuple

import torch

from oracle import hotpath as H
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic

# a reduced configuration (small tables) keeps CPU-side construction fast; structure is unchanged
SMALL = dict(log2_hashmap_size=15, num_levels=16, max_res=2048,
             proposal_net_args_list=[
                 {"hidden_dim": 16, "log2_hashmap_size": 13, "num_levels": 5, "max_res": 128, "use_linear": False},
                 {"hidden_dim": 16, "log2_hashmap_size": 13, "num_levels": 5, "max_res": 256, "use_linear": False}])


def oracle_config(cfg: ThermalNerfModelConfig) -> H.OracleConfig:
    return H.OracleConfig(
        num_levels=cfg.num_levels, base_res=cfg.base_res, max_res=cfg.max_res,
        log2_hashmap_size=cfg.log2_hashmap_size, features_per_level=cfg.features_per_level,
        hidden_dim=cfg.hidden_dim, appearance_embed_dim=cfg.appearance_embed_dim,
        proposal_net_args_list=[dict(a, base_res=16) for a in cfg.proposal_net_args_list],
        num_proposal_samples_per_ray=tuple(cfg.num_proposal_samples_per_ray),
        num_nerf_samples_per_ray=cfg.num_nerf_samples_per_ray, near_plane=cfg.near_plane, far_plane=cfg.far_plane,
        use_average_appearance_embedding=cfg.use_average_appearance_embedding,
        disable_scene_contraction=cfg.disable_scene_contraction, sh_input=cfg.sh_input,
        sh_grad=cfg.sh_direction_gradient, use_same_proposal_network=cfg.use_same_proposal_network,
        use_gradient_scaling=cfg.use_gradient_scaling, proposal_initial_sampler=cfg.proposal_initial_sampler,
        trunc_exp_clamp_min=cfg.trunc_exp_clamp_min,
    )


@functools.lru_cache(maxsize=8)
def build(kind: str = "stress", S: int = 48, small: bool = True, num_images: int = 8, **over):
    """Returns (cpu_model, cpu_state_dict, oracle_cfg).  Cached: treat the results as read-only."""
    kw = dict(SMALL) if small else {}
    if over.pop("one_proposal_network", False):  # (lru_cache needs hashable arguments: a flag instead of a list of dicts)
        nets = kw.get("proposal_net_args_list") or ThermalNerfModelConfig().proposal_net_args_list
        kw.update(use_same_proposal_network=True, proposal_net_args_list=[nets[-1]])
    kw.update(over)
    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=S, **kw)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=num_images)
    synthetic.fill_model_(model, kind)
    model.eval()
    return model, synthetic.model_state_dict_cpu(model), oracle_config(cfg)


def rays(h: int = 16, w: int = 16, view: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    o, d, _ = synthetic.orbit_camera_rays(h, w, view=view)
    return o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()


# ---- BASELINE config 1 analogue: 1 k optimisation steps of the CPU reference path on a small analytic scene --------------------
# (the reference's config 1 is "1k iters on the CPU PyTorch reference path: plumbing, no GPU"; its scene data does not exist in
# the checkout, so the closed-form RGB + thermal scene of thermo_nerf_amd.synthetic stands in.)  Sized so that one oracle step
# takes ~0.1 s on a few cores: 64-ray batches, (64, 32) proposal + 24 field samples, the SMALL tables.
CONFIG1 = dict(steps=1000, rays_per_batch=64, views=6, res=20, proposal=(64, 32), S=24, seed=5)


# the same problem at config 1's STATED step size [BASELINE.md §5; REF config_thermal_nerf.py:27]: 4096 rays per step, P = (256, 96),
# S = 48 (the nerfacto default), full-size tables — 30 steps, held step for step (tools/make_config1_golden.py --batch4096)
CONFIG1_FULL = dict(steps=30, rays_per_batch=4096, views=6, res=48, proposal=(256, 96), S=48, seed=6, small=False)


def config1_problem(c=None):
    """Model / oracle config + the ray pool, per-step batch indices and jitter draws shared by the CPU run and the HIP run."""
    c = c or CONFIG1
    cm, sd, ocfg = build("init", c["S"], small=c.get("small", True), camera_optimizer_mode="off",
                         num_proposal_samples_per_ray=c["proposal"], num_images=c["views"])
    o, d, cam = [], [], []
    for v in range(c["views"]):
        ov, dv, _ = synthetic.orbit_camera_rays(c["res"], c["res"], view=v, num_views=c["views"], elevation_deg=(-10.0, 20.0, 50.0)[v % 3])
        o.append(ov.reshape(-1, 3))
        d.append(dv.reshape(-1, 3))
        cam.append(torch.full((ov.shape[0] * ov.shape[1], 1), v, dtype=torch.long))
    o, d, cam = torch.cat(o), torch.cat(d), torch.cat(cam)
    img, th = synthetic.analytic_scene(o, d)
    g = torch.Generator().manual_seed(c["seed"])
    idx = torch.randint(0, o.shape[0], (c["steps"], c["rays_per_batch"]), generator=g)
    jit = torch.rand(c["steps"], 3, c["rays_per_batch"], 1, generator=g)
    # held-out PIXELS: training view 1 at 30 x 30 — none of its pixel centres coincides with one of the 20 x 20 training pixels,
    # every ray passes between them and 60 % of them hit the sphere.  (A held-out VIEW checks nothing at this size: 6 views x 400
    # pixels x 1000 steps fit the training rays (thermal MAE 0.25 -> 0.03) through density near each camera, and 6 degrees beside a
    # training view the sphere's thermal MAE is WORSE than at initialisation, 0.14 -> 0.28, for the CPU reference path itself.)
    ho, hd, _ = synthetic.orbit_camera_rays(30, 30, view=1, num_views=c["views"], elevation_deg=20.0)
    ho, hd = ho.reshape(-1, 3).contiguous(), hd.reshape(-1, 3).contiguous()
    himg, hth = synthetic.analytic_scene(ho, hd)
    return dict(model=cm, sd=sd, ocfg=ocfg, o=o, d=d, cam=cam, image=img, thermal=th, idx=idx, jitter=jit,
                held_out=dict(o=ho, d=hd, image=himg, thermal=hth))


def proposal_updates(steps: int):
    """Per step: does the sampler update its networks?  nerfstudio's ProposalNetworkSampler.step_cb / generate_ray_samples with
    the reference's schedule [REF thermal_nerf_model.py:152-161]."""
    from oracle import training as T

    since, flags = 0, []
    for step in range(steps):
        since += 1
        upd = since > T.update_schedule(step) or step < 10
        if upd:
            since = 0
        flags.append(upd)
    return flags


def config1_oracle_run(prob, steps=None, log=None, threads: int = 1, batch_order=None, dtype=torch.float32):
    """Adam (lr 1e-2, eps 1e-15 [REF config_thermal_nerf.py:32-45], no decay over this horizon) over torch autograd on the CPU
    oracle.  Returns (loss per step, final state dict).  ``threads`` / ``batch_order`` (a permutation of the batch positions):
    the SAME optimisation in another floating-point summation order — what separates two valid fp32 runs of the reference path
    (tools/make_config1_golden.py records several; the HIP run is held inside their envelope).  ``dtype=torch.float64``: the same
    optimisation in double precision — the yardstick that says which way an fp32 run leans while the trajectories still agree."""
    from oracle import training as T

    sd, ocfg = prob["sd"], prob["ocfg"]
    steps = steps or prob["idx"].shape[0]
    # ONE thread + deterministic algorithms: two runs on one host are bit-identical (the intra-op pool's reduction order moved the
    # late loss windows by 5x and made the CPU test a coin toss); as fast as 8 threads on these op sizes (74 s / 1000 steps)
    threads_before, det = torch.get_num_threads(), torch.are_deterministic_algorithms_enabled()
    dtype_before = torch.get_default_dtype()
    torch.set_num_threads(threads)
    torch.use_deterministic_algorithms(True)
    torch.set_default_dtype(dtype)  # (the oracle creates temporaries in the default dtype)

    def cast(t):
        return t.to(dtype) if t.is_floating_point() else t

    try:  # (an exception must not leave deterministic mode / a one-thread pool on for the rest of the pytest process)
        leaves = {k: cast(v).clone().requires_grad_(True) for k, v in sd.items()
                  if v.is_floating_point() and not k.endswith((".aabb", ".scalings")) and not k.startswith("camera_optimizer")}
        frozen = {k: cast(v) for k, v in sd.items() if k not in leaves}
        opt = torch.optim.Adam(list(leaves.values()), lr=1e-2, eps=1e-15)
        upd = proposal_updates(steps)
        losses = []
        for i in range(steps):
            ix = prob["idx"][i]
            if batch_order is not None:  # the same batch, its rays in another order: every batch reduction sums in another order
                ix = ix[batch_order]
            jitter = [cast(j if batch_order is None else j[batch_order]) for j in prob["jitter"][i]]
            batch = {"image": cast(prob["image"][ix]), "thermal": cast(prob["thermal"][ix])}
            out = H.get_outputs({**frozen, **leaves}, cast(prob["o"][ix]), cast(prob["d"][ix]), prob["cam"][ix], ocfg, training=True,
                                jitter=jitter, anneal=T.proposal_anneal(i), proposal_requires_grad=upd[i])
            loss = sum(T.get_loss_dict(out, batch, T.get_metrics_dict(out, batch, True), True).values())
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.item())
            if log and i % log == 0:
                print(i, losses[-1], flush=True)
    finally:
        torch.set_num_threads(threads_before)
        torch.use_deterministic_algorithms(det)
        torch.set_default_dtype(dtype_before)
    return losses, {**frozen, **{k: v.detach() for k, v in leaves.items()}}


def held_out_quality(prob, sd):
    """(RGB PSNR dB, thermal MAE in normalised units, thermal MAE over the rays that hit the sphere) of the oracle's eval render of
    the held-out rays."""
    h = prob["held_out"]
    with torch.no_grad():
        out = H.get_outputs(sd, h["o"], h["d"], None, prob["ocfg"])
    mse = ((out["rgb"] - h["image"]) ** 2).mean().item()
    err = (out["thermal"] - h["thermal"]).abs()
    hit = h["thermal"] != 0.15  # (the backdrop's temperature, synthetic.analytic_scene)
    return -10.0 * torch.log10(torch.tensor(mse)).item(), err.mean().item(), err[hit].mean().item()
