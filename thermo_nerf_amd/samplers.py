"""Ray samplers with nerfstudio's names (NS model_components.ray_samplers), built at
[REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:164-179] and run at :222-224 — on MI355X.

  UniformLinDispPiecewiseSampler  -> tn_sample_initial     (proposal_initial_sampler="piecewise")
  UniformSampler                  -> tn_sample_initial     (uniform_spacing=1; proposal_initial_sampler="uniform")
  PDFSampler                      -> tn_sample_pdf         (a11)
  ProposalNetworkSampler          -> the level loop of SURVEY A.7, same update/anneal bookkeeping
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _hip
from .rays import RayBundle, RaySamples

_const_cache: Dict[Tuple[str, int, str], Tensor] = {}


def linspace_bins(n: int, device) -> Tensor:
    """torch.linspace(0, 1, n+1) evaluated on the host exactly as nerfstudio does, then kept on the device."""
    key = ("lin", n, str(device))
    if key not in _const_cache:
        _const_cache[key] = torch.linspace(0.0, 1.0, n + 1).to(device)
    return _const_cache[key]


def pdf_positions(num_bins: int, device, training: bool) -> Tensor:
    """NS PDFSampler's ``u``: eval adds the half-bin offset; training adds the per-ray jitter in the kernel."""
    key = ("u_train" if training else "u_eval", num_bins, str(device))
    if key not in _const_cache:
        u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins)
        if not training:
            u = u + 1.0 / (2 * num_bins)
        _const_cache[key] = u.to(device)
    return _const_cache[key]


def _samples_from_bins(ray_bundle: RayBundle, spacing: Tensor, eucl: Tensor, uniform_spacing: bool = False) -> RaySamples:
    rs = ray_bundle.get_ray_samples(
        bin_starts=eucl[..., :-1, None], bin_ends=eucl[..., 1:, None],
        spacing_starts=spacing[..., :-1, None], spacing_ends=spacing[..., 1:, None],
        spacing_to_euclidean_fn=None,
    )
    rs.spacing_bins, rs.eucl_bins = spacing, eucl
    # stands in for nerfstudio's spacing_to_euclidean_fn closure: which spacing function the bins were made with, so that
    # PDFSampler maps its new bins the same way
    rs.uniform_spacing = uniform_spacing
    return rs


class LazyRaySamples(RaySamples):
    """The RaySamples of one level of a training step, built from the level's bin edges on first use.  The step itself only
    exchanges ``spacing_bins`` / ``eucl_bins`` (what the loss kernels read); the [R,n,1] views, the broadcast frustums and
    ``deltas`` (an elementwise launch) are made when something asks for them."""

    _LAZY = frozenset(("frustums", "camera_indices", "deltas", "spacing_starts", "spacing_ends", "spacing_to_euclidean_fn",
                       "metadata", "nears", "fars"))

    def __init__(self, ray_bundle: RayBundle, spacing: Tensor, eucl: Tensor, uniform_spacing: bool = False) -> None:
        d = self.__dict__
        d["_bundle"], d["spacing_bins"], d["eucl_bins"], d["uniform_spacing"] = ray_bundle, spacing, eucl, uniform_spacing

    def __getattribute__(self, name):
        if name in LazyRaySamples._LAZY:
            d = object.__getattribute__(self, "__dict__")
            if name not in d:
                rs = _samples_from_bins(d["_bundle"], d["spacing_bins"], d["eucl_bins"], d["uniform_spacing"])
                for k in LazyRaySamples._LAZY:
                    d.setdefault(k, getattr(rs, k))
            return d[name]
        return object.__getattribute__(self, name)

    def __repr__(self) -> str:
        return f"LazyRaySamples(spacing_bins={tuple(self.spacing_bins.shape)}, uniform_spacing={self.uniform_spacing})"


def draw_jitter(num_rays: int, counts: Sequence[int], single_jitter: bool, device) -> Tensor:
    """The stratified draws of one training forward, in the layout tn_render_inputs.jitter takes: single_jitter -> [3,R]
    (one draw per ray and level); otherwise the flat [R,P0+1] | [R,P1+1] | [R,S+1] of tn_render_config.per_sample_jitter
    (one draw per bin edge) [NS SpacedSampler / PDFSampler.generate_ray_samples: rand((R,1)) vs rand((R,n+1))]."""
    if single_jitter:
        return torch.rand((len(counts), num_rays), dtype=torch.float32, device=device)
    return torch.rand((num_rays * sum(n + 1 for n in counts),), dtype=torch.float32, device=device)


def jitter_levels(jitter, num_rays: int, counts: Sequence[int], single_jitter: bool) -> Tuple[Tensor, List[Tensor]]:
    """(flat device tensor, per-level views) of ``jitter``: a tensor in draw_jitter's layout, or a sequence of per-level
    tensors ([R] / [R,1] single, [R,n+1] per edge) as the modular samplers take them."""
    if not torch.is_tensor(jitter):
        jitter = torch.cat([j.reshape(-1).to(torch.float32) for j in jitter])
    flat = _hip.require_device_tensor(jitter.reshape(-1), "jitter")
    sizes = [num_rays * (1 if single_jitter else n + 1) for n in counts]
    if flat.numel() != sum(sizes):
        raise ValueError("jitter holds %d draws; single_jitter=%s over %d rays and levels %s takes %d"
                         % (flat.numel(), single_jitter, num_rays, tuple(counts), sum(sizes)))
    return flat, list(torch.split(flat, sizes))


class UniformLinDispPiecewiseSampler(nn.Module):
    """NS UniformLinDispPiecewiseSampler (the "piecewise" proposal_initial_sampler default)."""

    uniform_spacing = False

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False):
        super().__init__()
        self.num_samples = num_samples
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter

    def forward(self, ray_bundle: RayBundle, num_samples: Optional[int] = None,
                t_rand: Optional[Tensor] = None) -> RaySamples:
        n = num_samples or self.num_samples
        assert n is not None and ray_bundle.nears is not None and ray_bundle.fars is not None
        o = _hip.require_device_tensor(ray_bundle.origins, "origins")
        R = o.shape[0]
        draws = R if self.single_jitter else R * (n + 1)  # NS SpacedSampler: rand((R,1)) or rand((R,n+1))
        if self.train_stratified and self.training:
            if t_rand is None:
                t_rand = torch.rand((draws,), dtype=torch.float32, device=o.device)
            if t_rand.numel() != draws:
                raise ValueError("t_rand holds %d draws, single_jitter=%s takes %d" % (t_rand.numel(), self.single_jitter, draws))
        else:
            t_rand = None
        nears = _hip.require_device_tensor(ray_bundle.nears.reshape(-1), "nears")
        fars = _hip.require_device_tensor(ray_bundle.fars.reshape(-1), "fars")
        spacing = torch.empty((R, n + 1), dtype=torch.float32, device=o.device)
        eucl = torch.empty((R, n + 1), dtype=torch.float32, device=o.device)
        lib = _hip.load()
        _hip.check(
            lib.tn_sample_initial(linspace_bins(n, o.device).data_ptr(),
                                  None if t_rand is None else _hip.require_device_tensor(t_rand.reshape(-1), "t_rand").data_ptr(),
                                  nears.data_ptr(), fars.data_ptr(), R, n,
                                  int(self.uniform_spacing) | (0 if self.single_jitter else 2), spacing.data_ptr(),
                                  eucl.data_ptr(), _hip.current_stream()),
            "tn_sample_initial",
        )
        return _samples_from_bins(ray_bundle, spacing, eucl, self.uniform_spacing)


class UniformSampler(UniformLinDispPiecewiseSampler):
    """NS UniformSampler (proposal_initial_sampler="uniform", REF thermal_nerf_model.py:164-170): the same SpacedSampler
    with spacing_fn = spacing_fn_inv = identity, i.e. bins spread linearly in distance between near and far."""

    uniform_spacing = True


class PDFSampler(nn.Module):
    """NS PDFSampler(include_original=False, histogram_padding=0.01)."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = False, histogram_padding: float = 0.01) -> None:
        super().__init__()
        if include_original or histogram_padding != 0.01:
            raise NotImplementedError("kernels implement include_original=False, histogram_padding=0.01 "
                                      "(how ProposalNetworkSampler builds its PDFSampler)")
        self.num_samples = num_samples
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter

    def forward(self, ray_bundle: RayBundle, ray_samples: RaySamples, weights: Tensor,
                num_samples: Optional[int] = None, eps: float = 1e-5, u_rand: Optional[Tensor] = None) -> RaySamples:
        n_out = num_samples or self.num_samples
        assert n_out is not None and ray_samples.spacing_bins is not None
        w = _hip.require_device_tensor(weights[..., 0], "weights")
        R, n_in = w.shape
        jitter = self.train_stratified and self.training
        draws = R if self.single_jitter else R * (n_out + 1)  # NS PDFSampler: rand((R,1)) or rand((R,n_out+1))
        if jitter:
            if u_rand is None:
                u_rand = torch.rand((draws,), dtype=torch.float32, device=w.device)
            if u_rand.numel() != draws:
                raise ValueError("u_rand holds %d draws, single_jitter=%s takes %d" % (u_rand.numel(), self.single_jitter, draws))
        else:
            u_rand = None
        existing = _hip.require_device_tensor(ray_samples.spacing_bins, "spacing_bins")
        uniform = bool(getattr(ray_samples, "uniform_spacing", False))
        nears = _hip.require_device_tensor(ray_bundle.nears.reshape(-1), "nears")
        fars = _hip.require_device_tensor(ray_bundle.fars.reshape(-1), "fars")
        spacing = torch.empty((R, n_out + 1), dtype=torch.float32, device=w.device)
        eucl = torch.empty((R, n_out + 1), dtype=torch.float32, device=w.device)
        lib = _hip.load()
        _hip.check(
            lib.tn_sample_pdf(w.data_ptr(), existing.data_ptr(), pdf_positions(n_out + 1, w.device, jitter).data_ptr(),
                              None if u_rand is None else _hip.require_device_tensor(u_rand.reshape(-1), "u_rand").data_ptr(),
                              nears.data_ptr(), fars.data_ptr(), R, n_in, n_out,
                              int(uniform) | (0 if self.single_jitter else 2), spacing.data_ptr(),
                              eucl.data_ptr(), _hip.current_stream()),
            "tn_sample_pdf",
        )
        return _samples_from_bins(ray_bundle, spacing, eucl, uniform)


class ProposalNetworkSampler(nn.Module):
    """NS ProposalNetworkSampler: initial sampler (piecewise by default), then PDF resampling per level."""

    def __init__(self, num_proposal_samples_per_ray: Tuple[int, ...] = (64,), num_nerf_samples_per_ray: int = 32,
                 num_proposal_network_iterations: int = 2, single_jitter: bool = False,
                 update_sched: Callable = lambda x: 1, initial_sampler: Optional[nn.Module] = None,
                 pdf_sampler: Optional[PDFSampler] = None) -> None:
        super().__init__()
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        if self.num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        self.initial_sampler = initial_sampler or UniformLinDispPiecewiseSampler(single_jitter=single_jitter)
        self.pdf_sampler = pdf_sampler or PDFSampler(include_original=False, single_jitter=single_jitter)
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step: int) -> None:
        self._step = step
        self._steps_since_update += 1

    def forward(self, ray_bundle: RayBundle, density_fns: Sequence[Callable],
                jitter: Optional[Sequence[Tensor]] = None) -> Tuple[RaySamples, List[Tensor], List[RaySamples]]:
        weights_list: List[Tensor] = []
        ray_samples_list: List[RaySamples] = []
        n = self.num_proposal_network_iterations
        weights = None
        ray_samples = None
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        for i_level in range(n + 1):
            is_prop = i_level < n
            num_samples = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            jit = None if jitter is None else jitter[i_level]
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=num_samples, t_rand=jit)
            else:
                annealed = weights if self._anneal == 1.0 else torch.pow(weights, self._anneal)
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, annealed, num_samples=num_samples, u_rand=jit)
            if is_prop:
                density = density_fns[i_level](ray_samples.frustums.get_positions())
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        if updated:
            self._steps_since_update = 0
        assert ray_samples is not None
        return ray_samples, weights_list, ray_samples_list
