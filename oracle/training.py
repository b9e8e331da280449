"""TEST INFRASTRUCTURE — CPU oracle for the TRAINING step of the hot path.  NOT product code.

Restates, in differentiable plain torch, what one optimisation step of the reference evaluates on top of
``oracle.hotpath.get_outputs(training=True)``:

* ``get_loss_dict``                      [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:277-326]
* ``get_metrics_dict`` (psnr, distortion) NS NerfactoModel.get_metrics_dict, consumed at [REF :303-306]
* ``interlevel_loss`` / ``distortion_loss`` / ``lossfun_outer`` / ``outer`` / ``ray_samples_to_sdist``
                                          NS model_components/losses.py (nerfstudio 1.1.5, pinned by uv.lock)
* proposal-weight anneal + update schedule NS NerfactoModel.get_training_callbacks, built at [REF :152-161]
* ``exp_map_SO3xR3`` / ``apply_pose_adjustment``  NS cameras/lie_groups.py + CameraOptimizer.apply_to_raybundle (mode "SO3xR3",
                                          [REF nerfacto_config/thermal_nerfacto.py:38-40]), the call at [REF :218-219]

Gradients come from torch autograd over these functions; the HIP backward kernels are compared against them.
**Parity unpinned**: nerfstudio is not importable here and the reference holds no numeric vector for its losses.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from . import hotpath as H

EPS = 1.0e-7  # NS losses.EPS


def ray_samples_to_sdist(s: H.Samples) -> Tensor:
    """NS losses.ray_samples_to_sdist: [R, n+1] spacing-domain edges."""
    return torch.cat([s.spacing_starts[..., 0], s.spacing_ends[..., -1:, 0]], dim=-1)


def outer(t0_starts: Tensor, t0_ends: Tensor, t1_starts: Tensor, t1_ends: Tensor, y1: Tensor) -> Tensor:
    """NS losses.outer: upper envelope of histogram (t1, y1) on the intervals (t0)."""
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1
    idx_lo = torch.clamp(idx_lo, min=0, max=y1.shape[-1] - 1)
    idx_hi = torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right")
    idx_hi = torch.clamp(idx_hi, min=0, max=y1.shape[-1] - 1)
    cy1_lo = torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)
    cy1_hi = torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1)
    return cy1_hi - cy1_lo


def lossfun_outer(t: Tensor, w: Tensor, t_env: Tensor, w_env: Tensor) -> Tensor:
    """NS losses.lossfun_outer (mip-NeRF 360 eq. 13)."""
    w_outer = outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + EPS)


def interlevel_loss(weights_list: Sequence[Tensor], samples_list: Sequence[H.Samples]) -> Tensor:
    """NS losses.interlevel_loss: the final level is detached, each proposal level is pulled above it."""
    c = ray_samples_to_sdist(samples_list[-1]).detach()
    w = weights_list[-1][..., 0].detach()
    loss = 0.0
    for s, weights in zip(samples_list[:-1], weights_list[:-1]):
        cp = ray_samples_to_sdist(s)
        wp = weights[..., 0]
        loss = loss + torch.mean(lossfun_outer(c, w, cp, wp))
    return loss


def lossfun_distortion(t: Tensor, w: Tensor) -> Tensor:
    """NS losses.lossfun_distortion (mip-NeRF 360 eq. 15), O(S^2) form."""
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    loss_inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    loss_intra = torch.sum(w**2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return loss_inter + loss_intra


def distortion_loss(weights_list: Sequence[Tensor], samples_list: Sequence[H.Samples]) -> Tensor:
    """NS losses.distortion_loss on the final level."""
    c = ray_samples_to_sdist(samples_list[-1])
    w = weights_list[-1][..., 0]
    return torch.mean(lossfun_distortion(c, w))


def get_metrics_dict(outputs: Dict, batch: Dict, training: bool) -> Dict[str, Tensor]:
    """NS NerfactoModel.get_metrics_dict: psnr always, distortion in training."""
    m = {"psnr": H.psnr(outputs["rgb"], batch["image"])}
    if training:
        m["distortion"] = distortion_loss(outputs["weights_list"], outputs["ray_samples_list"])
    return m


def get_loss_dict(outputs: Dict, batch: Dict, metrics_dict: Optional[Dict], training: bool,
                  interlevel_loss_mult: float = 1.0, distortion_loss_mult: float = 0.002,
                  pass_rgb_gradients: bool = True, pass_thermal_gradients: bool = True) -> Dict[str, Tensor]:
    """[REF thermal_nerf_model.py:277-326], no predicted normals.  background "last_sample" leaves 3-channel ground truth
    untouched (NS blend_background_for_loss_computation).  ``thermal_loss_weight`` exists in the reference config
    [REF :53-54] but is never applied [REF :319-323]; the field's pass_rgb_gradients / pass_thermal_gradients flags gate the
    two image losses [REF :295, :321].  Key order and values are pinned by the reference's own method (fixture G7)."""
    loss = {}
    if pass_rgb_gradients:
        loss["rgb_loss"] = torch.nn.functional.mse_loss(batch["image"], outputs["rgb"])  # [REF :294-295]
    if training:
        loss["interlevel_loss"] = interlevel_loss_mult * interlevel_loss(outputs["weights_list"], outputs["ray_samples_list"])
        assert metrics_dict is not None and "distortion" in metrics_dict  # [REF :301]
        loss["distortion_loss"] = distortion_loss_mult * metrics_dict["distortion"]
    if pass_thermal_gradients:
        loss["thermal"] = torch.nn.functional.mse_loss(outputs["thermal"], batch["thermal"])  # [REF :319-323]
    return loss


def proposal_anneal(step: int, max_num_iters: int = 1000, slope: float = 10.0) -> float:
    """NS NerfactoModel.get_training_callbacks.set_anneal (use_proposal_weight_anneal=True)."""
    train_frac = float(np.clip(step / max_num_iters, 0, 1))
    return slope * train_frac / ((slope - 1) * train_frac + 1)


def update_schedule(step: int, proposal_warmup: int = 5000, proposal_update_every: int = 5) -> float:
    """[REF thermal_nerf_model.py:152-161]."""
    return float(np.clip(np.interp(step, [0, proposal_warmup], [0, proposal_update_every]), 1, proposal_update_every))


def exp_map_SO3xR3(tangent_vector: Tensor) -> Tensor:
    """NS cameras/lie_groups.exp_map_SO3xR3: [N,6] = (translation, log-rotation) -> [N,3,4] = [R | t], Rodrigues with the
    squared angle clamped at 1e-4."""
    log_rot = tangent_vector[:, 3:]
    nrms = (log_rot * log_rot).sum(1)
    rot_angles = torch.clamp(nrms, 1e-4).sqrt()
    rot_angles_inv = 1.0 / rot_angles
    fac1 = rot_angles_inv * rot_angles.sin()
    fac2 = rot_angles_inv * rot_angles_inv * (1.0 - rot_angles.cos())
    skews = torch.zeros((log_rot.shape[0], 3, 3), dtype=log_rot.dtype)
    skews[:, 0, 1] = -log_rot[:, 2]
    skews[:, 0, 2] = log_rot[:, 1]
    skews[:, 1, 0] = log_rot[:, 2]
    skews[:, 1, 2] = -log_rot[:, 0]
    skews[:, 2, 0] = -log_rot[:, 1]
    skews[:, 2, 1] = log_rot[:, 0]
    skews_square = torch.bmm(skews, skews)
    ret = torch.zeros(tangent_vector.shape[0], 3, 4, dtype=tangent_vector.dtype)
    ret[:, :3, :3] = fac1[:, None, None] * skews + fac2[:, None, None] * skews_square + torch.eye(3, dtype=log_rot.dtype)[None]
    ret[:, :3, 3] = tangent_vector[:, :3]
    return ret


def exp_map_SE3(tangent_vector: Tensor) -> Tensor:
    """NS cameras/lie_groups.exp_map_SE3: [N,6] = (translation part u, log-rotation w) -> [N,3,4] = [R(w) | V(w) u] with
    theta^2 clamped at 1e-4 as in exp_map_SO3xR3; V = I + fac2 K + fac3 K^2, fac3 = (theta - sin theta) / theta^3."""
    log_rot = tangent_vector[:, 3:]
    nrms = (log_rot * log_rot).sum(1)
    rot_angles = torch.clamp(nrms, 1e-4).sqrt()
    rot_angles_inv = 1.0 / rot_angles
    fac1 = rot_angles_inv * rot_angles.sin()
    fac2 = rot_angles_inv * rot_angles_inv * (1.0 - rot_angles.cos())
    fac3 = rot_angles_inv * rot_angles_inv * rot_angles_inv * (rot_angles - rot_angles.sin())
    skews = torch.zeros((log_rot.shape[0], 3, 3), dtype=log_rot.dtype)
    skews[:, 0, 1] = -log_rot[:, 2]
    skews[:, 0, 2] = log_rot[:, 1]
    skews[:, 1, 0] = log_rot[:, 2]
    skews[:, 1, 2] = -log_rot[:, 0]
    skews[:, 2, 0] = -log_rot[:, 1]
    skews[:, 2, 1] = log_rot[:, 0]
    skews_square = torch.bmm(skews, skews)
    eye = torch.eye(3, dtype=log_rot.dtype)[None]
    ret = torch.zeros(tangent_vector.shape[0], 3, 4, dtype=tangent_vector.dtype)
    ret[:, :3, :3] = fac1[:, None, None] * skews + fac2[:, None, None] * skews_square + eye
    v = fac2[:, None, None] * skews + fac3[:, None, None] * skews_square + eye
    ret[:, :3, 3] = torch.bmm(v, tangent_vector[:, :3, None])[:, :, 0]
    return ret


def apply_pose_adjustment(pose_adjustment: Tensor, camera_indices: Tensor, origins: Tensor, directions: Tensor, mode: str = "SO3xR3"):
    """NS CameraOptimizer.forward (gather the rays' rows, exponentiate: mode "SO3xR3" or "SE3") + apply_to_raybundle."""
    exp = exp_map_SE3 if mode == "SE3" else exp_map_SO3xR3
    m = exp(pose_adjustment[camera_indices.reshape(-1).long(), :])
    return origins + m[:, :3, 3], torch.bmm(m[:, :3, :3], directions[..., None]).squeeze(-1)


def loss_and_grads(sd: Dict[str, Tensor], origins: Tensor, directions: Tensor, camera_indices: Tensor, batch: Dict,
                   cfg: H.OracleConfig, jitter: Sequence[Tensor], anneal: float = 1.0,
                   proposal_requires_grad: bool = True, interlevel_loss_mult: float = 1.0,
                   distortion_loss_mult: float = 0.002, dtype=torch.float32):
    """One training forward + backward on the CPU: returns (outputs, loss_dict, {param name: grad}).
    ``dtype=torch.float64`` gives a tighter yardstick for the fp32 kernels' gradients."""
    # hotpath creates fp32 temporaries; run it in the requested default dtype
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        leaves = {}
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith((".aabb", ".scalings")):
                leaves[k] = v.detach().to(dtype).clone().requires_grad_(True)
        sdd = {k: (leaves[k] if k in leaves else (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd.items()}
        jit = [j.to(dtype) for j in jitter]
        origins = origins.detach().to(dtype).clone().requires_grad_(True)      # what a camera optimizer differentiates
        directions = directions.detach().to(dtype).clone().requires_grad_(True)
        out = H.get_outputs(sdd, origins, directions, camera_indices, cfg, training=True,
                            jitter=jit, anneal=anneal, proposal_requires_grad=proposal_requires_grad)
        b = {k: v.to(dtype) for k, v in batch.items()}
        metrics = get_metrics_dict(out, b, True)
        loss_dict = get_loss_dict(out, b, metrics, True, interlevel_loss_mult, distortion_loss_mult)
        total = sum(loss_dict.values())
        total.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
        grads["__origins__"], grads["__directions__"] = origins.grad, directions.grad
    finally:
        torch.set_default_dtype(prev)
    return out, loss_dict, grads
