"""TEST INFRASTRUCTURE — CPU oracle for ThermoNeRF's volumetric-rendering hot path.  NOT product code.

What this is
------------
A plain-torch fp32, op-for-op restatement of what ``ThermalNerfModel.get_outputs``
[REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:210-275] executes when the locked environment
(nerfstudio==1.1.5 [REF uv.lock:2743-2744], no tinycudann) falls back to nerfstudio's pure-PyTorch
``HashEncoding`` / ``MLP`` / samplers / renderers.  Each function cites the reference line (``REF``) or
the nerfstudio 1.1.5 symbol (``NS``; third-party, pinned in uv.lock, source not vendored in
/root/reference) it follows.

PARITY PINNING STATUS
---------------------
* Pinned against the real reference code (run in the build container by tools/make_golden.py, fixtures under
  tests/golden/, checked in tests/test_oracle_golden.py): ``render_thermal`` (G1: thermal_renderer.py), ``mae_thermal``
  (G2: thermal_metrics.py), ``render_rgb`` (G4: rgb_concat/rgbt_renderer.py), the thermal head (G5:
  thermal_field_head.py), the wiring of ``field_outputs`` (G6: thermal_field.py run on oracle-built nerfstudio stand-ins)
  and the orchestration / loss dictionary of ``get_outputs`` + ``oracle.training.get_loss_dict`` (G7:
  thermal_nerf_model.py:210-326 run the same way).  G6 / G7 pin WIRING (which module sees which tensor, keys, gates,
  multipliers), not nerfstudio's arithmetic.
* **Parity unpinned** at the nerfstudio boundary: the reference's own tests hold no numeric vector for
  this path and nerfstudio cannot be imported here.  Everything tagged ``NS`` below
  restates nerfstudio 1.1.5's published torch-fallback algorithm.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

# ----------------------------------------------------------------------------------------------
# configuration (defaults = NerfactoModelConfig 1.1.5 defaults consumed at
# [REF thermal_nerf_model.py:91-184])
# ----------------------------------------------------------------------------------------------


@dataclass
class OracleConfig:
    num_levels: int = 16
    base_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2
    hidden_dim: int = 64
    geo_feat_dim: int = 15
    appearance_embed_dim: int = 32
    proposal_net_args_list: List[dict] = field(
        default_factory=lambda: [
            {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "base_res": 16},
            {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "base_res": 16},
        ]
    )
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    near_plane: float = 0.05
    far_plane: float = 1000.0
    use_average_appearance_embedding: bool = True
    disable_scene_contraction: bool = False
    # [REF thermal_field.py:86] passes 1.0 positionally; proposal nets keep the 1.0 default.
    average_init_density: float = 1.0
    # SURVEY Appendix A.6 [UNSURE]: torch-fallback SHEncoding receives (d+1)/2 unmodified.
    sh_input: str = "shifted"  # "shifted" | "unit"
    # NS SHEncoding.pytorch_fwd is decorated @torch.no_grad() — in the torch fallback the SH basis is a
    # constant of the graph and no gradient reaches the ray directions through it.  True = differentiate through the basis
    # (what a tcnn SH encoding does); the product mirrors this switch as config.sh_direction_gradient.
    sh_grad: bool = False
    use_same_proposal_network: bool = False  # [REF thermal_nerf_model.py:127-139]: one network for every proposal level
    use_gradient_scaling: bool = False  # [REF :228-231]: NS scale_gradients_by_distance_squared on the field outputs
    proposal_initial_sampler: str = "piecewise"  # [REF :164-170]: "uniform" -> NS UniformSampler
    # NS field_components/activations.py trunc_exp (taken from torch-ngp): backward g * exp(clamp(x, min=-15, max=15)).
    # SURVEY A.3 tags the bounds [UNSURE]: -inf here gives the upper-clamp-only form.  Differs only for raw densities below
    # -15 (gradient factor < 3e-7).  The product mirrors it as config.trunc_exp_clamp_min.
    trunc_exp_clamp_min: float = -15.0


# ----------------------------------------------------------------------------------------------
# NS HashEncoding (torch path)
# ----------------------------------------------------------------------------------------------

_PRIMES = (1, 2654435761, 805459861)


def hash_scalings(num_levels: int, min_res: int, max_res: int) -> Tensor:
    """NS HashEncoding.__init__: ``scalings = floor(min_res * growth ** arange(L))`` evaluated by torch
    in float32 with a numpy-float64 growth factor."""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * growth**levels)


def hash_fn(coords: Tensor, table_size: int, hash_offset: Tensor) -> Tensor:
    """NS HashEncoding.hash_fn: int32 coords [..., L, 3] * int64 primes, xor, mod T, + level offset."""
    c = coords * torch.tensor(_PRIMES, dtype=torch.int64)
    x = torch.bitwise_xor(c[..., 0], c[..., 1])
    x = torch.bitwise_xor(x, c[..., 2])
    x %= table_size
    x += hash_offset
    return x


def hash_encode(p: Tensor, table: Tensor, scalings: Tensor, log2_hashmap_size: int) -> Tensor:
    """NS HashEncoding.pytorch_fwd. ``p`` [N,3] in [0,1]; ``table`` [L*T, F]; returns [N, L*F]."""
    L = scalings.shape[0]
    T = 2**log2_hashmap_size
    hash_offset = torch.arange(L) * T
    x = p[..., None, :]
    scaled = x * scalings.view(-1, 1)
    sc = torch.ceil(scaled).type(torch.int32)
    sf = torch.floor(scaled).type(torch.int32)
    off = scaled - sf

    def h(ix, iy, iz):
        return hash_fn(torch.cat([ix[..., 0:1], iy[..., 1:2], iz[..., 2:3]], dim=-1), T, hash_offset)

    h0 = h(sc, sc, sc)
    h1 = h(sc, sf, sc)
    h2 = h(sf, sf, sc)
    h3 = h(sf, sc, sc)
    h4 = h(sc, sc, sf)
    h5 = h(sc, sf, sf)
    h6 = h(sf, sf, sf)
    h7 = h(sf, sc, sf)
    f0, f1, f2, f3 = table[h0], table[h1], table[h2], table[h3]
    f4, f5, f6, f7 = table[h4], table[h5], table[h6], table[h7]
    ox, oy, oz = off[..., 0:1], off[..., 1:2], off[..., 2:3]
    f03 = f0 * ox + f3 * (1 - ox)
    f12 = f1 * ox + f2 * (1 - ox)
    f56 = f5 * ox + f6 * (1 - ox)
    f47 = f4 * ox + f7 * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    enc = f0312 * oz + f4756 * (1 - oz)
    return torch.flatten(enc, start_dim=-2, end_dim=-1)


# ----------------------------------------------------------------------------------------------
# NS MLP (torch path), SceneContraction, SH
# ----------------------------------------------------------------------------------------------


def mlp(x: Tensor, layers: Sequence[Tuple[Tensor, Tensor]], out_activation: Optional[str]) -> Tensor:
    """NS MLP.pytorch_fwd: Linear(+bias), ReLU between layers, optional out activation."""
    for i, (w, b) in enumerate(layers):
        x = torch.nn.functional.linear(x, w, b)
        if i < len(layers) - 1:
            x = torch.relu(x)
    if out_activation == "sigmoid":
        x = torch.sigmoid(x)
    elif out_activation is not None:
        raise ValueError(out_activation)
    return x


def _layers(sd: Dict[str, Tensor], prefix: str, n: int) -> List[Tuple[Tensor, Tensor]]:
    return [(sd[f"{prefix}.layers.{i}.weight"], sd[f"{prefix}.layers.{i}.bias"]) for i in range(n)]


def contract_inf(x: Tensor) -> Tensor:
    """NS SceneContraction(order=inf).forward, built at [REF thermal_nerf_model.py:94]."""
    mag = torch.linalg.norm(x, ord=float("inf"), dim=-1)[..., None]
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


def normalized_positions(pos: Tensor, cfg: OracleConfig, aabb: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """NS NerfactoField.get_density / HashMLPDensityField.get_density position normalisation + selector."""
    if not cfg.disable_scene_contraction:
        p = contract_inf(pos)
        p = (p + 2.0) / 4.0
    else:
        assert aabb is not None
        p = (pos - aabb[0]) / (aabb[1] - aabb[0])  # NS SceneBox.get_normalized_positions
    selector = ((p > 0.0) & (p < 1.0)).all(dim=-1)
    p = p * selector[..., None]
    return p, selector


def sh4(d: Tensor) -> Tensor:
    """NS components_from_spherical_harmonics, 16 components."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x**2, y**2, z**2
    c = torch.zeros((*d.shape[:-1], 16), dtype=d.dtype)
    c[..., 0] = 0.28209479177387814
    c[..., 1] = 0.4886025119029199 * y
    c[..., 2] = 0.4886025119029199 * z
    c[..., 3] = 0.4886025119029199 * x
    c[..., 4] = 1.0925484305920792 * x * y
    c[..., 5] = 1.0925484305920792 * y * z
    c[..., 6] = 0.9461746957575601 * zz - 0.31539156525251999
    c[..., 7] = 1.0925484305920792 * x * z
    c[..., 8] = 0.5462742152960396 * (xx - yy)
    c[..., 9] = 0.5900435899266435 * y * (3 * xx - yy)
    c[..., 10] = 2.890611442640554 * x * y * z
    c[..., 11] = 0.4570457994644658 * y * (5 * zz - 1)
    c[..., 12] = 0.3731763325901154 * z * (5 * zz - 3)
    c[..., 13] = 0.4570457994644658 * x * (5 * zz - 1)
    c[..., 14] = 1.445305721320277 * z * (xx - yy)
    c[..., 15] = 0.5900435899266435 * x * (xx - 3 * yy)
    return c


# ----------------------------------------------------------------------------------------------
# fields
# ----------------------------------------------------------------------------------------------


class _TruncExp(torch.autograd.Function):
    """NS field_components.activations.trunc_exp: forward exp(x), backward g * exp(clamp(x, min=lo, max=15)); lo = -15 in
    nerfstudio (from torch-ngp), -inf = clamp from above only (OracleConfig.trunc_exp_clamp_min)."""

    @staticmethod
    def forward(ctx, x, lo):
        ctx.save_for_backward(x)
        ctx.lo = lo
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(torch.clamp(x, min=ctx.lo, max=15.0)), None


def trunc_exp(x: Tensor, lo: float = -15.0) -> Tensor:
    return _TruncExp.apply(x, float(lo))


class _GradientScaler(torch.autograd.Function):
    """NS model_components.losses._GradientScaler: identity forward, gradient times ``scaling`` backward."""

    @staticmethod
    def forward(ctx, value, scaling):
        ctx.save_for_backward(scaling)
        return value, scaling

    @staticmethod
    def backward(ctx, output_grads, grad_scaling):
        (scaling,) = ctx.saved_tensors
        return output_grads * scaling, grad_scaling


def scale_gradients_by_distance_squared(outputs: Dict[str, Tensor], starts: Tensor, ends: Tensor) -> Dict[str, Tensor]:
    """NS losses.scale_gradients_by_distance_squared, applied at [REF thermal_nerf_model.py:228-231] to EVERY field output."""
    ray_dist = (starts + ends) / 2
    scaling = torch.square(ray_dist).clamp(0, 1)
    return {k: _GradientScaler.apply(v, scaling)[0] for k, v in outputs.items()}


def proposal_density(sd: Dict[str, Tensor], level: int, positions: Tensor, cfg: OracleConfig) -> Tensor:
    """NS HashMLPDensityField.density_fn/get_density (a5), built at [REF thermal_nerf_model.py:136-149].
    positions [...,3] -> density [...,1]."""
    pre = f"proposal_networks.{0 if cfg.use_same_proposal_network else level}"
    args = cfg.proposal_net_args_list[min(level, len(cfg.proposal_net_args_list) - 1)]
    shape = positions.shape[:-1]
    p, selector = normalized_positions(positions, cfg, sd.get(f"{pre}.aabb"))
    enc = hash_encode(
        p.view(-1, 3), sd[f"{pre}.mlp_base.encoder.hash_table"], sd[f"{pre}.mlp_base.encoder.scalings"],
        args["log2_hashmap_size"],
    )
    raw = mlp(enc, _layers(sd, f"{pre}.mlp_base.mlp", 2), None).view(*shape, -1)
    density = 1.0 * trunc_exp(raw, cfg.trunc_exp_clamp_min)  # average_init_density default 1.0
    return density * selector[..., None]


def field_density(sd: Dict[str, Tensor], positions: Tensor, cfg: OracleConfig) -> Tuple[Tensor, Tensor]:
    """NS NerfactoField.get_density (a8) as called from [REF thermal_field.py:186-190]."""
    shape = positions.shape[:-1]
    p, selector = normalized_positions(positions, cfg, sd.get("field.aabb"))
    enc = hash_encode(
        p.view(-1, 3), sd["field.mlp_base.encoder.hash_table"], sd["field.mlp_base.encoder.scalings"],
        cfg.log2_hashmap_size,
    )
    h = mlp(enc, _layers(sd, "field.mlp_base.mlp", 2), None).view(*shape, -1)
    raw, geo = torch.split(h, [1, cfg.geo_feat_dim], dim=-1)
    density = cfg.average_init_density * trunc_exp(raw, cfg.trunc_exp_clamp_min)
    density = density * selector[..., None]
    return density, geo


def field_outputs(
    sd: Dict[str, Tensor], directions: Tensor, geo: Tensor, camera_indices: Optional[Tensor],
    cfg: OracleConfig, training: bool,
) -> Tuple[Tensor, Tensor]:
    """ThermalNerfactoTField.get_outputs [REF thermal_field.py:108-181]: returns (rgb[...,3], thermal[...,1])."""
    shape = directions.shape[:-1]
    d = (directions + 1.0) / 2.0  # NS get_normalized_directions [REF :117]
    d_flat = d.reshape(-1, 3)
    enc_in = d_flat if cfg.sh_input == "shifted" else directions.reshape(-1, 3)
    if cfg.sh_grad:
        d_enc = sh4(enc_in)  # [REF :119]
    else:
        with torch.no_grad():  # NS SHEncoding.pytorch_fwd runs under @torch.no_grad()
            d_enc = sh4(enc_in)  # [REF :119]
    emb = sd["field.embedding_appearance.embedding.weight"]
    if training:
        app = emb[camera_indices.reshape(-1)]  # [REF :124-125]
    elif cfg.use_average_appearance_embedding:
        app = torch.ones((d_flat.shape[0], cfg.appearance_embed_dim)) * emb.mean(dim=0)  # [REF :128-132]
    else:
        app = torch.zeros((d_flat.shape[0], cfg.appearance_embed_dim))  # [REF :133-137]
    geo_flat = geo.reshape(-1, cfg.geo_feat_dim)
    h = torch.cat([d_enc, geo_flat, app], dim=-1)  # [REF :160-167]
    rgb = mlp(h, _layers(sd, "field.mlp_head", 3), "sigmoid").view(*shape, -1)  # [REF :168]
    t_hidden = mlp(geo_flat, _layers(sd, "field.mlp_thermal", 2), "sigmoid")  # [REF :90-98,175-177]
    thermal = torch.nn.functional.linear(
        t_hidden, sd["field.field_head_thermal.net.weight"], sd["field.field_head_thermal.net.bias"]
    ).view(*shape, -1)  # [REF :178; thermal_field_head.py:50-51,66] no activation
    return rgb, thermal


# ----------------------------------------------------------------------------------------------
# samplers (NS ray_samplers.py)
# ----------------------------------------------------------------------------------------------


def spacing_fn(x: Tensor) -> Tensor:
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))


def spacing_fn_inv(x: Tensor) -> Tensor:
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))


@dataclass
class Samples:
    """Minimal stand-in for NS RaySamples: all [R, n, 1] except s_near/s_far [R,1]."""

    starts: Tensor
    ends: Tensor
    spacing_starts: Tensor
    spacing_ends: Tensor
    s_near: Tensor
    s_far: Tensor
    uniform: bool = False  # NS UniformSampler: spacing_fn = spacing_fn_inv = identity

    @property
    def deltas(self) -> Tensor:
        return self.ends - self.starts

    def to_euclidean(self, x: Tensor) -> Tensor:
        u = x * self.s_far + (1 - x) * self.s_near
        return u if self.uniform else spacing_fn_inv(u)


def _samples_from_bins(bins: Tensor, s_near: Tensor, s_far: Tensor, uniform: bool = False) -> Samples:
    eucl = bins * s_far + (1 - bins) * s_near
    if not uniform:
        eucl = spacing_fn_inv(eucl)
    n_rays = eucl.shape[0]
    sb = bins.expand(n_rays, -1)
    return Samples(eucl[..., :-1, None], eucl[..., 1:, None], sb[..., :-1, None], sb[..., 1:, None], s_near, s_far,
                   uniform)


def sample_initial(nears: Tensor, fars: Tensor, num_samples: int, t_rand: Optional[Tensor],
                   uniform: bool = False) -> Samples:
    """NS UniformLinDispPiecewiseSampler / SpacedSampler.generate_ray_samples (a4); ``uniform`` = NS UniformSampler
    (proposal_initial_sampler="uniform", REF thermal_nerf_model.py:164-170: identity spacing functions).
    ``t_rand`` = the stratified draw in training: [R,1] under single_jitter (one per ray), [R,n+1] otherwise (one per bin
    edge; NS SpacedSampler draws rand((R, n+1)) and the same expression broadcasts); None in eval."""
    bins = torch.linspace(0.0, 1.0, num_samples + 1)[None, ...]
    if t_rand is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    s_near, s_far = (nears, fars) if uniform else (spacing_fn(nears), spacing_fn(fars))
    return _samples_from_bins(bins, s_near, s_far, uniform)


def pdf_u(num_bins: int) -> Tensor:
    """The eval-mode sample positions of NS PDFSampler (host-side constant, also fed to the HIP path)."""
    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins)
    return u + 1.0 / (2 * num_bins)


def sample_pdf(prev: Samples, weights: Tensor, num_samples: int, rand: Optional[Tensor]) -> Samples:
    """NS PDFSampler.generate_ray_samples (a11); histogram_padding 0.01, eps 1e-5.  ``rand``: [R,1] under single_jitter,
    [R,num_samples+1] otherwise (NS draws rand((R, num_samples+1)); ``u + rand / num_bins`` broadcasts either)."""
    num_bins = num_samples + 1
    w = weights[..., 0] + 0.01
    ws = torch.sum(w, dim=-1, keepdim=True)
    padding = torch.relu(1e-5 - ws)
    w = w + padding / w.shape[-1]
    ws = ws + padding
    pdf = w / ws
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    if rand is not None:
        u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins)
        u = u.expand(size=(*cdf.shape[:-1], num_bins))
        u = u + rand / num_bins
    else:
        u = pdf_u(num_bins).expand(size=(*cdf.shape[:-1], num_bins))
    u = u.contiguous()
    existing = torch.cat([prev.spacing_starts[..., 0], prev.spacing_ends[..., -1:, 0]], dim=-1)
    inds = torch.searchsorted(cdf, u, side="right")
    below = torch.clamp(inds - 1, 0, existing.shape[-1] - 1)
    above = torch.clamp(inds, 0, existing.shape[-1] - 1)
    cdf_g0 = torch.gather(cdf, -1, below)
    bins_g0 = torch.gather(existing, -1, below)
    cdf_g1 = torch.gather(cdf, -1, above)
    bins_g1 = torch.gather(existing, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
    bins = bins_g0 + t * (bins_g1 - bins_g0)
    bins = bins.detach()  # NS PDFSampler: "Stop gradients" — sample positions never carry gradient
    return _samples_from_bins(bins, prev.s_near, prev.s_far, prev.uniform)


def get_weights(deltas: Tensor, densities: Tensor) -> Tensor:
    """NS RaySamples.get_weights (a6), called at [REF thermal_nerf_model.py:233]."""
    dd = deltas * densities
    alphas = 1 - torch.exp(-dd)
    trans = torch.cumsum(dd[..., :-1, :], dim=-2)
    trans = torch.cat([torch.zeros((*trans.shape[:1], 1, 1)), trans], dim=-2)
    trans = torch.exp(-trans)
    return torch.nan_to_num(alphas * trans)


# ----------------------------------------------------------------------------------------------
# renderers
# ----------------------------------------------------------------------------------------------


def render_rgb(rgb: Tensor, weights: Tensor, training: bool) -> Tensor:
    """NS RGBRenderer(background_color="last_sample") (a13); witness fork at
    [REF thermo_nerf/rgb_concat/rgbt_renderer.py:62-81,159-174]."""
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = torch.sum(weights * rgb, dim=-2)
    acc = torch.sum(weights, dim=-2)
    comp = comp + rgb[..., -1, :] * (1.0 - acc)
    if not training:
        comp = torch.clamp(comp, min=0.0, max=1.0)
    return comp


def render_thermal(thermal: Tensor, weights: Tensor, training: bool) -> Tensor:
    """ThermalRenderer.forward + combine_thermal [REF thermal_renderer.py:113-149, 27-80]:
    background is forced to "last_sample" (:49)."""
    if not training:
        thermal = torch.nan_to_num(thermal)  # [REF :136-137]
    comp = torch.sum(weights * thermal, dim=-2)  # [REF :55]
    acc = torch.sum(weights, dim=-2)  # [REF :56]
    comp = comp + thermal[..., -1, :] * (1.0 - acc)  # [REF :68-70,79]
    if not training:
        comp = torch.clamp(comp, min=0.0, max=1.0)  # [REF :146-147]
    return comp


def render_accumulation(weights: Tensor) -> Tensor:
    return torch.sum(weights, dim=-2)


def render_depth_median(weights: Tensor, starts: Tensor, ends: Tensor) -> Tensor:
    """NS DepthRenderer(method="median") (a13)."""
    steps = (starts + ends) / 2
    cw = torch.cumsum(weights[..., 0], dim=-1)
    split = torch.ones((*weights.shape[:-2], 1)) * 0.5
    idx = torch.searchsorted(cw, split, side="left")
    idx = torch.clamp(idx, 0, steps.shape[-2] - 1)
    return torch.gather(steps[..., 0], dim=-1, index=idx)


class Outputs(dict):
    """get_outputs' dict (keys and order = the reference's, fixture G7) + a diagnostic that is NOT part of the reference path:
    ``median_ties[key]`` for "depth" / "prop_depth_i" — how close the cumulative weight comes to the 0.5 split next to the median
    index, and the neighbouring steps.  tests/test_gpu_parity.py::check_outputs accepts a median-depth mismatch only on such a tie."""

    median_ties: Dict[str, Dict[str, Tensor]]


def median_depth_ties(weights: Tensor, starts: Tensor, ends: Tensor) -> Dict[str, Tensor]:
    """For render_depth_median's index i (first cw[i] >= 0.5, clamped): ``margin_above`` = |cw[i] - 0.5| (a path whose rounding
    leaves cw[i] just below the split answers steps[i+1] = ``above``), ``margin_below`` = |cw[i-1] - 0.5| (one whose cw[i-1]
    just reaches it answers steps[i-1] = ``below``).  All [R,1]."""
    steps = ((starts + ends) / 2)[..., 0]
    cw = torch.cumsum(weights[..., 0], dim=-1)
    n = steps.shape[-1]
    idx = torch.clamp(torch.searchsorted(cw, torch.full((*cw.shape[:-1], 1), 0.5), side="left"), 0, n - 1)
    lo, hi = (idx - 1).clamp_min(0), (idx + 1).clamp_max(n - 1)
    return {"margin_above": (torch.gather(cw, -1, idx) - 0.5).abs(), "margin_below": (torch.gather(cw, -1, lo) - 0.5).abs(),
            "above": torch.gather(steps, -1, hi), "below": torch.gather(steps, -1, lo)}


def render_depth_expected(weights: Tensor, starts: Tensor, ends: Tensor) -> Tensor:
    """NS DepthRenderer(method="expected"): clip uses the call-global min/max of steps (a13)."""
    steps = (starts + ends) / 2
    depth = torch.sum(weights * steps, dim=-2) / (torch.sum(weights, -2) + 1e-10)
    return torch.clip(depth, steps.min(), steps.max())


# ----------------------------------------------------------------------------------------------
# the model forward
# ----------------------------------------------------------------------------------------------


def collider(origins: Tensor, cfg: OracleConfig, training: bool) -> Tuple[Tensor, Tensor]:
    """NS NearFarCollider.set_nears_and_fars (a3): eval resets the near plane to 0."""
    ones = torch.ones_like(origins[..., 0:1])
    near = cfg.near_plane if training else 0.0
    return ones * near, ones * cfg.far_plane


def positions_of(origins: Tensor, directions: Tensor, s: Samples) -> Tensor:
    """NS Frustums.get_positions: origins + directions * (starts + ends) / 2."""
    return origins[:, None, :] + directions[:, None, :] * (s.starts + s.ends) / 2


def proposal_sampler(
    sd: Dict[str, Tensor], origins: Tensor, directions: Tensor, nears: Tensor, fars: Tensor,
    cfg: OracleConfig, jitter: Optional[Sequence[Tensor]] = None, anneal: float = 1.0,
    proposal_requires_grad: bool = True,
) -> Tuple[Samples, List[Tensor], List[Samples]]:
    """NS ProposalNetworkSampler.generate_ray_samples, invoked at [REF thermal_nerf_model.py:222-224].
    ``proposal_requires_grad`` = NS's ``updated`` flag (steps_since_update > update_sched(step) or step < 10):
    when False the proposal densities are evaluated under no_grad."""
    weights_list: List[Tensor] = []
    samples_list: List[Samples] = []
    n = len(cfg.num_proposal_samples_per_ray)
    weights = None
    s = None
    for lvl in range(n + 1):
        is_prop = lvl < n
        num = cfg.num_proposal_samples_per_ray[lvl] if is_prop else cfg.num_nerf_samples_per_ray
        jit = None if jitter is None else jitter[lvl]
        if lvl == 0:
            s = sample_initial(nears, fars, num, jit, uniform=cfg.proposal_initial_sampler == "uniform")
        else:
            s = sample_pdf(s, torch.pow(weights, anneal), num, jit)
        if is_prop:
            if proposal_requires_grad:
                density = proposal_density(sd, lvl, positions_of(origins, directions, s), cfg)
            else:
                with torch.no_grad():
                    density = proposal_density(sd, lvl, positions_of(origins, directions, s), cfg)
            weights = get_weights(s.deltas, density)
            weights_list.append(weights)
            samples_list.append(s)
    return s, weights_list, samples_list


def get_outputs(
    sd: Dict[str, Tensor], origins: Tensor, directions: Tensor, camera_indices: Optional[Tensor],
    cfg: OracleConfig, training: bool = False, jitter: Optional[Sequence[Tensor]] = None,
    anneal: float = 1.0, return_intermediates: bool = False, proposal_requires_grad: bool = True,
) -> Dict[str, Tensor]:
    """Model.forward (collider) + ThermalNerfModel.get_outputs [REF thermal_nerf_model.py:210-275].
    All inputs CPU fp32: origins/directions [R,3]; camera_indices [R,1] int64 (training only)."""
    nears, fars = collider(origins, cfg, training)
    s, weights_list, samples_list = proposal_sampler(sd, origins, directions, nears, fars, cfg, jitter, anneal,
                                                     proposal_requires_grad)
    pos = positions_of(origins, directions, s)
    density, geo = field_density(sd, pos, cfg)  # [REF thermal_field.py:186-190]
    dirs = directions[:, None, :].expand(-1, pos.shape[1], -1)
    cam = None if camera_indices is None else camera_indices[:, None, :].expand(-1, pos.shape[1], -1)
    rgb_s, thermal_s = field_outputs(sd, dirs, geo, cam, cfg, training)
    if cfg.use_gradient_scaling:  # [REF :228-231]
        scaled = scale_gradients_by_distance_squared({"rgb": rgb_s, "thermal": thermal_s, "density": density}, s.starts, s.ends)
        rgb_s, thermal_s, density = scaled["rgb"], scaled["thermal"], scaled["density"]
    weights = get_weights(s.deltas, density)  # [REF :233]
    weights_list.append(weights)
    samples_list.append(s)
    out = {
        "rgb": render_rgb(rgb_s, weights, training),  # [REF :237]
        "accumulation": render_accumulation(weights),  # [REF :243]
        "depth": render_depth_median(weights, s.starts, s.ends),  # [REF :238-239]
        "expected_depth": render_depth_expected(weights, s.starts, s.ends),  # [REF :240-242]
    }
    if training:  # [REF :262-265] (key order as in the reference's dict: pinned by fixture G7)
        out["weights_list"] = weights_list
        out["ray_samples_list"] = samples_list
    for i in range(len(cfg.num_proposal_samples_per_ray)):  # [REF :267-270]
        out[f"prop_depth_{i}"] = render_depth_median(weights_list[i], samples_list[i].starts, samples_list[i].ends)
    out["thermal"] = render_thermal(thermal_s, weights, training)  # [REF :271-273]
    out = Outputs(out)
    with torch.no_grad():  # (test diagnostic, see Outputs)
        out.median_ties = {"depth": median_depth_ties(weights, s.starts, s.ends)}
        for i in range(len(cfg.num_proposal_samples_per_ray)):
            out.median_ties[f"prop_depth_{i}"] = median_depth_ties(weights_list[i], samples_list[i].starts, samples_list[i].ends)
    if return_intermediates:
        out.setdefault("weights_list", weights_list)
        out.setdefault("ray_samples_list", samples_list)
        out["density"] = density
        out["geo"] = geo
        out["rgb_samples"] = rgb_s
        out["thermal_samples"] = thermal_s
    return out


def get_outputs_for_camera_ray_bundle(
    sd: Dict[str, Tensor], origins: Tensor, directions: Tensor, cfg: OracleConfig, chunk: int,
) -> Dict[str, Tensor]:
    """NS Model.get_outputs_for_camera_ray_bundle (a14): origins/directions [H,W,3], row-major chunks."""
    H, W = origins.shape[:2]
    o, d = origins.reshape(-1, 3), directions.reshape(-1, 3)
    outs: Dict[str, List[Tensor]] = {}
    ties: Dict[str, Dict[str, List[Tensor]]] = {}
    with torch.no_grad():
        for i in range(0, H * W, chunk):
            r = get_outputs(sd, o[i : i + chunk], d[i : i + chunk], None, cfg, training=False)
            for k, v in r.items():
                if isinstance(v, Tensor):
                    outs.setdefault(k, []).append(v)
            for k, t in r.median_ties.items():
                for name, v in t.items():
                    ties.setdefault(k, {}).setdefault(name, []).append(v)
    res = Outputs({k: torch.cat(v).view(H, W, -1) for k, v in outs.items()})
    res.median_ties = {k: {name: torch.cat(v).view(H, W, -1) for name, v in t.items()} for k, t in ties.items()}
    return res


# ----------------------------------------------------------------------------------------------
# metrics
# ----------------------------------------------------------------------------------------------


def mae_thermal(gt: Tensor, pred: Tensor, cold_flag: bool, max_temperature: float, min_temperature: float,
                threshold: Optional[float] = None) -> Tensor:
    """[REF thermo_nerf/thermal_nerf/thermal_metrics.py:5-34]."""
    if threshold:
        sel = (gt < threshold) if cold_flag else (gt > threshold)
        gt, pred = gt[sel], pred[sel]
    span = max_temperature - min_temperature
    return torch.mean(torch.abs((gt * span + min_temperature) - (pred * span + min_temperature)))


def psnr(pred: Tensor, gt: Tensor) -> Tensor:
    """torchmetrics PeakSignalNoiseRatio(data_range=1.0) [REF thermal_nerf_model.py:200]."""
    return 10.0 * torch.log10(1.0 / torch.mean((pred - gt) ** 2))
