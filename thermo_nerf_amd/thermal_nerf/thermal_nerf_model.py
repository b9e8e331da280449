"""ThermalNerfModel — ThermoNeRF's model with its forward pass on MI355X.

Interface mirror of [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:46-275]: same config class/fields,
constructor (``metadata`` must hold the "thermal" key), module names (``field``, ``proposal_networks``,
``proposal_sampler``, ``collider``, ``renderer_*``, ``thermal_renderer``, ``camera_optimizer``) and the same
``get_outputs`` output dictionary.

Two execution forms of ``get_outputs`` (both HIP only, no PyTorch arithmetic on the path):
  * ``config.fused=True``  — one C-ABI call, ``tn_render_rays_fwd`` (proposal kernel + main-field kernel);
  * ``config.fused=False`` — the reference's own call sequence (sampler -> field.forward -> get_weights ->
    renderers), every step one HIP entry point; used for Field-level callers and as the on-GPU cross-check.
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Type

import numpy as np
import torch
from torch import Tensor, nn

from .. import _hip
from ..fields import FieldHeadNames, HashMLPDensityField
from ..nerfacto_config.thermal_nerfacto import KERNEL_FAMILY, ThermalNerfactoModel, ThermalNerfactoModelConfig
from ..rays import RayBundle
from ..rendered_image_modalities import RenderedImageModality
from ..renderers import AccumulationRenderer, DepthRenderer, RGBRenderer
from ..samplers import (ProposalNetworkSampler, UniformSampler, draw_jitter, jitter_levels, linspace_bins, pdf_positions,
                        _samples_from_bins)
from ..scene import NearFarCollider, SceneBox, SceneContraction
from .thermal_field import ThermalNerfactoTField
from .thermal_field_head import FieldHeadNamesT
from .thermal_renderer import ThermalRenderer


_ENGINES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


@dataclass
class ThermalNerfModelConfig(ThermalNerfactoModelConfig):
    """[REF thermal_nerf_model.py:46-56]"""

    _target: Type = field(default_factory=lambda: ThermalNerfModel)
    use_transient_embedding: bool = False
    thermal_loss_weight: float = 1.0
    pass_thermal_gradients: bool = True


class ThermalNerfModel(ThermalNerfactoModel):
    config: ThermalNerfModelConfig

    def __init__(self, config: ThermalNerfModelConfig, metadata: dict, scene_box: SceneBox, num_train_data: int,
                 **kwargs) -> None:
        if RenderedImageModality.THERMAL.value not in metadata.keys():  # REF :75-76
            raise ValueError("Thermal images not found in metadata.")
        super().__init__(config, scene_box, num_train_data, **kwargs)
        self.config = config
        self.max_temperature = config.max_temperature
        self.min_temperature = config.min_temperature
        self._workspace: Optional[Tensor] = None
        self._struct_key = None
        self._structs = None

    # ------------------------------------------------------------------------------------------------
    def populate_modules(self) -> None:
        """[REF thermal_nerf_model.py:86-208]"""
        cfg = self.config
        if cfg.implementation not in ("hip", "tcnn", "torch"):
            raise ValueError(cfg.implementation)
        scene_contraction = None if cfg.disable_scene_contraction else SceneContraction(order=float("inf"))
        budget = int(cfg.dense_grid_budget_mb) << 20

        self.field = ThermalNerfactoTField(
            self.scene_box.aabb, hidden_dim=cfg.hidden_dim, num_levels=cfg.num_levels, max_res=cfg.max_res,
            base_res=cfg.base_res, features_per_level=cfg.features_per_level,
            log2_hashmap_size=cfg.log2_hashmap_size, hidden_dim_color=cfg.hidden_dim_color,
            hidden_dim_transient=cfg.hidden_dim_transient, spatial_distortion=scene_contraction,
            num_images=self.num_train_data, use_pred_normals=cfg.predict_normals,
            use_average_appearance_embedding=cfg.use_average_appearance_embedding,
            appearance_embedding_dim=cfg.appearance_embed_dim, implementation="hip",
            use_transient_embedding=cfg.use_transient_embedding, pass_thermal_gradients=cfg.pass_thermal_gradients,
            sh_input=cfg.sh_input,
        )
        self.field.dense_budget_bytes = int(cfg.field_dense_grid_budget_mb) << 20
        self.camera_optimizer = cfg.camera_optimizer.setup(num_cameras=self.num_train_data, device="cpu")

        self.density_fns = []
        num_prop_nets = cfg.num_proposal_iterations
        self.proposal_networks = nn.ModuleList()
        if cfg.use_same_proposal_network:
            assert len(cfg.proposal_net_args_list) == 1, "Only one proposal network is allowed."
            network = HashMLPDensityField(self.scene_box.aabb, spatial_distortion=scene_contraction,
                                          **cfg.proposal_net_args_list[0], implementation="hip")
            network.dense_budget_bytes = budget
            self.proposal_networks.append(network)
            self.density_fns.extend([network.density_fn for _ in range(num_prop_nets)])
        else:
            for i in range(num_prop_nets):
                args = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
                network = HashMLPDensityField(self.scene_box.aabb, spatial_distortion=scene_contraction, **args,
                                              implementation="hip")
                network.dense_budget_bytes = budget
                self.proposal_networks.append(network)
            self.density_fns.extend([network.density_fn for network in self.proposal_networks])

        def update_schedule(step):  # REF :152-161
            return np.clip(np.interp(step, [0, cfg.proposal_warmup], [0, cfg.proposal_update_every]), 1,
                           cfg.proposal_update_every)

        # REF :164-170: "uniform" -> UniformSampler, anything else -> UniformLinDispPiecewiseSampler
        initial_sampler = None
        if cfg.proposal_initial_sampler == "uniform":
            initial_sampler = UniformSampler(single_jitter=cfg.use_single_jitter)
        self.proposal_sampler = ProposalNetworkSampler(
            num_nerf_samples_per_ray=cfg.num_nerf_samples_per_ray,
            num_proposal_samples_per_ray=cfg.num_proposal_samples_per_ray,
            num_proposal_network_iterations=cfg.num_proposal_iterations, single_jitter=cfg.use_single_jitter,
            update_sched=update_schedule, initial_sampler=initial_sampler,
        )
        self.collider = NearFarCollider(near_plane=cfg.near_plane, far_plane=cfg.far_plane)
        self.renderer_rgb = RGBRenderer(background_color=cfg.background_color)
        self.renderer_accumulation = AccumulationRenderer()
        self.renderer_depth = DepthRenderer(method="median")
        self.renderer_expected_depth = DepthRenderer(method="expected")
        self.thermal_renderer = ThermalRenderer()
        self.step = 0

    # ------------------------------------------------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """NS NerfactoModel.get_param_groups (+ camera_opt)."""
        groups = {
            "proposal_networks": list(self.proposal_networks.parameters()),
            "fields": list(self.field.parameters()),
        }
        cam = list(self.camera_optimizer.parameters())
        if cam:
            groups["camera_opt"] = cam
        return groups

    def set_step(self, step: int) -> None:
        """The two nerfstudio training callbacks: proposal-weight anneal and the sampler step."""
        self.step = step
        cfg = self.config
        if cfg.use_proposal_weight_anneal:
            n = cfg.proposal_weights_anneal_max_num_iters
            frac = float(np.clip(step / n, 0, 1))
            slope = cfg.proposal_weights_anneal_slope
            self.proposal_sampler.set_anneal(slope * frac / ((slope - 1) * frac + 1))
        self.proposal_sampler.step_cb(step)

    # ------------------------------------------------------------------------------------------------
    def get_outputs(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """[REF thermal_nerf_model.py:210-275]"""
        if self.config.predict_normals:
            # What the REFERENCE does with predict_normals=True, recorded by executing its own code (G9, tests/golden/
            # predict_normals.json, tools/make_golden_g9.py): its field override [REF thermal_field.py:108-181] never evaluates
            # nerfstudio's predicted-normals head, so get_outputs ends in KeyError(FieldHeadNames.PRED_NORMALS) at [REF :256-258]
            # in every forward, train or eval.  The switch belongs to the config surface, not to the reference's working
            # behaviour: the same exception, instead of outputs the reference cannot produce.
            raise KeyError(FieldHeadNames.PRED_NORMALS)
        if self.training:
            self.camera_optimizer.apply_to_raybundle(ray_bundle)  # REF :218-219
        if ray_bundle.nears is None or ray_bundle.fars is None:
            raise ValueError("ray_bundle.nears/fars are unset: call the model (forward applies the collider)")
        fusable = self._fusable()
        if self.training and torch.is_grad_enabled():
            from ..training import get_outputs_train  # taped forward: the outputs carry the HIP backward
            self.invalidate_prepared()
            return get_outputs_train(self, ray_bundle)
        if fusable:
            return self._get_outputs_fused(ray_bundle)
        return self._get_outputs_modular(ray_bundle)

    def _fusable(self) -> bool:
        cfg = self.config
        # use_gradient_scaling only rescales gradients (backward); use_same_proposal_network hands one network to both levels
        # (field.staged: MLP widths other than the reference's 64 run one launch per module / layer, thermal_field.py)
        return cfg.fused and cfg.num_proposal_iterations == 2 and not cfg.predict_normals and not self.field.staged

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """NS Model.get_outputs_for_camera_ray_bundle [REF render/renderer.py:182-187 calls it per camera].  In eval mode
        with the fused kernels the chunks (``eval_num_rays_per_chunk`` rays each, expected depth clipped per chunk exactly
        like the reference's per-chunk forward) are written straight into the [H*W,C] outputs by ``RayRenderEngine`` on two
        alternating HIP streams; otherwise the generic per-chunk loop runs."""
        if self.training or not self._fusable():
            return super().get_outputs_for_camera_ray_bundle(camera_ray_bundle)
        from ..engine import RayRenderEngine

        chunk = int(self.config.eval_num_rays_per_chunk)
        eng = _ENGINES.get(self)  # side table: streams / ctypes structs must not ride along in deepcopy / state_dict
        if eng is None or eng.chunk != chunk or eng.rc.early_stop_transmittance != float(self.config.early_termination_eps):
            eng = _ENGINES[self] = RayRenderEngine(weakref.proxy(self), chunk=chunk)  # proxy: the table must not keep the model alive
        h, w = camera_ray_bundle.origins.shape[:2]
        o = camera_ray_bundle.origins.reshape(-1, 3).to(self.device)
        d = camera_ray_bundle.directions.reshape(-1, 3).to(self.device)
        eng.rc.pdf_anneal = float(self.proposal_sampler._anneal)
        nears, fars = camera_ray_bundle.nears, camera_ray_bundle.fars
        if nears is not None and fars is not None:  # planes already on the bundle win over the collider's constants
            nears, fars = nears.reshape(-1).to(self.device), fars.reshape(-1).to(self.device)
        else:
            nears = fars = None
        out = eng.render(o, d, nears=nears, fars=fars)
        return {k: v.view(h, w, -1) for k, v in out.items()}

    # ------------------------------------------------------------------------------------------------
    def get_metrics_dict(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """NS NerfactoModel.get_metrics_dict: psnr, and the distortion metric that get_loss_dict scales
        [REF thermal_nerf_model.py:301-304]."""
        from ..training import distortion_loss

        gt_rgb = batch["image"].to(self.device)[..., :3]
        fused = self._fused_image_losses(outputs, batch)
        if fused is not None:
            metrics = {"psnr": fused[2]}
        else:
            metrics = {"psnr": 10.0 * torch.log10(1.0 / torch.mean((outputs["rgb"].detach() - gt_rgb) ** 2))}
        if self.training:
            metrics["distortion"] = distortion_loss(outputs["weights_list"], outputs["ray_samples_list"],
                                                    mult=self.config.distortion_loss_mult)
        return metrics

    def _fused_image_losses(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor]):
        """(rgb MSE, thermal MSE, PSNR) of a training batch from ONE kernel (tn_image_losses), shared by get_metrics_dict and
        get_loss_dict of the same outputs; None when the batch is not the plain training case (then torch's ops run)."""
        rgb, th = outputs.get("rgb"), outputs.get(RenderedImageModality.THERMAL.value)
        img, tgt = batch.get("image"), batch.get(RenderedImageModality.THERMAL.value)
        if not (self.training and torch.is_grad_enabled() and isinstance(rgb, Tensor) and rgb.is_cuda and rgb.dim() == 2
                and isinstance(th, Tensor) and isinstance(img, Tensor) and isinstance(tgt, Tensor) and img.shape == rgb.shape
                and tgt.numel() == th.numel() and rgb.dtype == torch.float32 and img.dtype == torch.float32
                and tgt.dtype == torch.float32):
            return None
        hit = self.__dict__.get("_image_loss_cache")
        if hit is not None and hit[0] is rgb and hit[1] is img and hit[2] is tgt:
            return hit[3]
        from ..training import image_losses

        res = image_losses(rgb, th, img.to(rgb.device), tgt.to(rgb.device))
        self.__dict__["_image_loss_cache"] = (rgb, img, tgt, res)
        return res

    def get_loss_dict(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor],
                      metrics_dict: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
        """[REF thermal_nerf_model.py:277-326]"""
        from ..training import interlevel_loss

        loss_dict: Dict[str, Tensor] = {}
        image = batch["image"].to(self.device)
        pred_rgb, gt_rgb = self.renderer_rgb.blend_background_for_loss_computation(
            pred_image=outputs["rgb"], pred_accumulation=outputs[RenderedImageModality.ACCUMULATION.value], gt_image=image)
        fused = self._fused_image_losses(outputs, batch) if pred_rgb is outputs["rgb"] and image.shape[-1] == 3 else None
        if self.field.pass_rgb_gradients:
            loss_dict["rgb_loss"] = fused[0] if fused is not None else torch.nn.functional.mse_loss(gt_rgb, pred_rgb)  # REF :294-295
        if self.training:
            loss_dict["interlevel_loss"] = interlevel_loss(outputs["weights_list"], outputs["ray_samples_list"],
                                                           mult=self.config.interlevel_loss_mult)  # REF :296-300
            assert metrics_dict is not None and "distortion" in metrics_dict  # REF :301
            term = getattr(metrics_dict["distortion"], "scaled_term", None)  # the kernel's own mult * metric, when it made one
            loss_dict["distortion_loss"] = term[1] if term is not None and term[0] == self.config.distortion_loss_mult \
                else self.config.distortion_loss_mult * metrics_dict["distortion"]
            if self.config.predict_normals:  # (unreachable through get_outputs, which raised: see there)
                raise KeyError(FieldHeadNames.PRED_NORMALS)
        thermal_batch = batch[RenderedImageModality.THERMAL.value].to(self.device)
        if self.field.pass_thermal_gradients:
            # the reference gates the thermal LOSS (not only the geo gradient) on this flag [REF :319-323]
            loss_dict[RenderedImageModality.THERMAL.value] = fused[1] if fused is not None else torch.nn.functional.mse_loss(
                outputs[RenderedImageModality.THERMAL.value], thermal_batch)
        self.__dict__.pop("_image_loss_cache", None)  # one (metrics, losses) pair per forward
        return loss_dict

    def get_image_metrics_and_images(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor],
                                     threshold: Optional[float] = None) -> Tuple[Dict[str, float], Dict[str, Tensor]]:
        """[REF thermal_nerf_model.py:328-393] on top of NS NerfactoModel.get_image_metrics_and_images: per-frame metrics
        and the images an evaluation saves, from the [H,W,C] outputs of ``get_outputs_for_camera_ray_bundle``.

        Metrics: ``psnr`` / ``ssim`` (RGB), ``psnr_thermal`` / ``ssim_thermal``, ``mae_thermal`` / ``mae_thermal_foreground``
        in degrees — PSNR as torchmetrics' PeakSignalNoiseRatio(data_range=1), SSIM as its
        structural_similarity_index_measure defaults (``tn_ssim_fwd``).  ``lpips`` / ``lpips_thermal`` are NaN: LPIPS is a
        pretrained AlexNet whose weights torchmetrics downloads; none exist offline.
        Images: ``img`` (ground truth | prediction), ``thermal`` and ``thermal_combined`` (NS "gray" colormap = the value on
        three channels), ``accumulation``, ``depth``, ``prop_depth_i`` — the last three as GREY maps with nerfstudio's depth
        normalisation and accumulation blend, where nerfstudio applies matplotlib's "turbo" lookup table (not available here)."""
        from ..metrics import psnr, ssim
        from .thermal_metrics import mae_thermal

        dev = outputs["rgb"].device
        gt_rgb = self.renderer_rgb.blend_background(batch["image"].to(dev))
        rgb, acc = outputs["rgb"], outputs[RenderedImageModality.ACCUMULATION.value]
        th = outputs[RenderedImageModality.THERMAL.value]
        gt_th = batch[RenderedImageModality.THERMAL.value].to(dev)

        def grey(x: Tensor) -> Tensor:  # NS colormaps.apply_float_colormap(x, "gray")
            return torch.nan_to_num(x, 0).repeat(1, 1, 3)

        def depth_map(d: Tensor) -> Tensor:  # NS colormaps.apply_depth_colormap without the lookup table
            near, far = float(torch.min(d)), float(torch.max(d))
            g = grey(torch.clip((d - near) / (far - near + 1e-10), 0, 1))
            return g * acc + (1 - acc)

        images = {
            "img": torch.cat([gt_rgb, rgb], dim=1),
            "accumulation": grey(acc),
            "depth": depth_map(outputs[RenderedImageModality.DEPTH.value]),
            RenderedImageModality.THERMAL.value: grey(th),
            RenderedImageModality.THERMAL_COMBINED.value: torch.cat([grey(gt_th), grey(th)], dim=1),
        }
        for i in range(self.config.num_proposal_iterations):
            key = f"prop_depth_{i}"
            if key in outputs:
                images[key] = depth_map(outputs[key])
        gt4, th4 = torch.moveaxis(gt_th, -1, 0)[None, ...], torch.moveaxis(th, -1, 0)[None, ...]  # [1,C,H,W] as the reference
        metrics = {
            "psnr": float(psnr(gt_rgb, rgb)),
            "ssim": float(ssim(rgb, gt_rgb)),
            "lpips": float("nan"),
            "psnr_thermal": float(psnr(gt_th, th)),
            "ssim_thermal": float(ssim(th, gt_th)),
            "lpips_thermal": float("nan"),
            "mae_thermal_foreground": float(mae_thermal(gt4, th4, self.config.cold, self.max_temperature,
                                                        self.min_temperature, threshold=threshold)),
            "mae_thermal": float(mae_thermal(gt4, th4, self.config.cold, self.max_temperature, self.min_temperature,
                                             threshold=None)),
        }
        return metrics, images

    # --- the reference's call sequence, one HIP entry point per nerfstudio module ----------------------
    def _get_outputs_modular(self, ray_bundle: RayBundle, jitter: Optional[Sequence[Tensor]] = None) -> Dict[str, Tensor]:
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(ray_bundle, density_fns=self.density_fns,
                                                                            jitter=jitter)
        field_outputs = self.field.forward(ray_samples, compute_normals=self.config.predict_normals)
        # REF :228-231 use_gradient_scaling: scale_gradients_by_distance_squared is the identity in the forward pass
        weights = ray_samples.get_weights(field_outputs[FieldHeadNames.DENSITY])
        weights_list.append(weights)
        ray_samples_list.append(ray_samples)
        rgb = self.renderer_rgb(rgb=field_outputs[FieldHeadNames.RGB], weights=weights)
        depth = self.renderer_depth(weights=weights, ray_samples=ray_samples)
        expected_depth = self.renderer_expected_depth(weights=weights, ray_samples=ray_samples)
        accumulation = self.renderer_accumulation(weights=weights)
        outputs = {
            "rgb": rgb,
            RenderedImageModality.ACCUMULATION.value: accumulation,
            RenderedImageModality.DEPTH.value: depth,
            "expected_depth": expected_depth,
        }
        if self.training:
            outputs["weights_list"] = weights_list
            outputs["ray_samples_list"] = ray_samples_list
        for i in range(self.config.num_proposal_iterations):
            outputs[f"prop_depth_{i}"] = self.renderer_depth(weights=weights_list[i], ray_samples=ray_samples_list[i])
        outputs[RenderedImageModality.THERMAL.value] = self.thermal_renderer(field_outputs[FieldHeadNamesT.THERMAL], weights)
        return outputs

    def train(self, mode: bool = True):
        """Every switch between training and evaluation drops the derived weight copies: a fused optimizer may have stepped
        since they were built (it does not bump Parameter._version), and the eval kernels must never see a stale blob or
        dense re-layout."""
        self.invalidate_prepared()
        _hip.join_pending()  # config.deferred_table_update: a table update still on the side streams ends here
        return super().train(mode)

    def state_dict(self, *args, **kwargs):
        _hip.join_pending()  # (the same: the copy a checkpoint takes runs on the calling stream)
        return super().state_dict(*args, **kwargs)

    def invalidate_prepared(self) -> None:
        """Drop every derived copy of the weights (MFMA blobs, dense re-layouts, cached C structs).  The caches key on
        parameter versions, but fused multi-tensor optimizers (torch.optim.Adam(fused=True)) update parameters without
        bumping them — so every training forward, after which weights are about to change, calls this."""
        self._struct_key = None
        self.field._prepared_key = None
        mods = self.__dict__.get("_tn_dense_modules")
        if mods is None:  # the module tree is fixed after populate_modules: walk it once, not every training step
            mods = self.__dict__["_tn_dense_modules"] = [m for m in self.modules() if hasattr(m, "_dense_key")]
        for mod in mods:
            mod._dense_key = None

    def named_parameter_lists(self) -> Tuple[List[str], List[nn.Parameter]]:
        """(names, parameters) in ``named_parameters()`` order, walked once (the training step passes every parameter to its
        autograd Function each iteration; the Parameter objects of a built model do not change, only their values)."""
        hit = self.__dict__.get("_tn_named_params")
        if hit is None:
            named = list(self.named_parameters())
            hit = self.__dict__["_tn_named_params"] = ([n for n, _ in named], [p for _, p in named])
        return hit

    # --- fused: one C-ABI call ------------------------------------------------------------------------
    def _c_structs(self):
        _hip.join_pending()  # eval launches read the tables on the calling stream
        # (the cached parameter list: walking the module tree costs more than a small eval call's kernels)
        key = tuple([(p.data_ptr(), p._version) for p in self.named_parameter_lists()[1]]) + (self.config.use_mfma,
                                                                                              self.config.mlp_precision)
        if self._struct_key != key:
            nets = len(self.proposal_networks)  # 1 with use_same_proposal_network [REF :127-139]
            self._structs = (self.proposal_networks[0].c_struct(), self.proposal_networks[min(1, nets - 1)].c_struct(),
                             self.field.c_struct(prepare=self.config.use_mfma, precision=self.config.mlp_precision))
            self._struct_key = key
        return self._structs

    def _get_outputs_fused(self, ray_bundle: RayBundle, jitter: Optional[Tensor] = None,
                           want_samples: Optional[bool] = None) -> Dict[str, Tensor]:
        lib = _hip.load()
        cfg = self.config
        training = bool(self.training)
        want_samples = training if want_samples is None else want_samples
        o = _hip.require_device_tensor(ray_bundle.origins, "origins")
        d = _hip.require_device_tensor(ray_bundle.directions, "directions")
        if o.dim() != 2:
            raise ValueError("get_outputs expects a flat ray bundle: origins [R,3]")
        R = o.shape[0]
        dev = o.device
        nears = _hip.require_device_tensor(ray_bundle.nears.reshape(-1), "nears")
        fars = _hip.require_device_tensor(ray_bundle.fars.reshape(-1), "fars")
        P0, P1 = cfg.num_proposal_samples_per_ray
        S = cfg.num_nerf_samples_per_ray
        prop0, prop1, fld = self._c_structs()

        rc = _hip.tn_render_config()
        rc.num_proposal_samples[0], rc.num_proposal_samples[1] = P0, P1
        rc.num_nerf_samples = S
        rc.training = 1 if training else 0
        rc.pdf_anneal = float(self.proposal_sampler._anneal)
        rc.early_stop_transmittance = 0.0 if training else float(cfg.early_termination_eps)
        rc.kernel_family = KERNEL_FAMILY[cfg.kernel_family]
        rc.initial_sampler = int(self.proposal_sampler.initial_sampler.uniform_spacing)
        rc.sample_split = int(getattr(cfg, "sample_split", 0))
        rc.per_sample_jitter = 0 if cfg.use_single_jitter else 1

        ins = _hip.tn_render_inputs()
        ins.origins, ins.directions, ins.nears, ins.fars = o.data_ptr(), d.data_ptr(), nears.data_ptr(), fars.data_ptr()
        cam = None
        if training:
            if ray_bundle.camera_indices is None:
                raise AttributeError("Camera indices are not provided.")
            cam = _hip.require_device_tensor(ray_bundle.camera_indices.reshape(-1).to(torch.int32), "camera_indices",
                                             torch.int32)
            if jitter is None:
                jitter = draw_jitter(R, (P0, P1, S), bool(cfg.use_single_jitter), dev)
            jitter = jitter_levels(jitter, R, (P0, P1, S), bool(cfg.use_single_jitter))[0]
        ins.camera_indices = _hip.ptr(cam)
        ins.jitter = _hip.ptr(jitter) if training else None
        ins.lin_bins0 = linspace_bins(P0, dev).data_ptr()
        ins.u1 = pdf_positions(P1 + 1, dev, training).data_ptr()
        ins.u2 = pdf_positions(S + 1, dev, training).data_ptr()

        buf = torch.empty((9, R), dtype=torch.float32, device=dev)  # 36 B/ray of outputs, one allocation
        rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        outs = _hip.tn_render_outputs()
        outs.rgb = rgb.data_ptr()
        names = ["accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1", "thermal"]
        for i, nm in enumerate(names):
            setattr(outs, nm, buf[i].data_ptr())
        extra = {}
        if want_samples:
            ns = (P0, P1, S)
            extra["w"] = [torch.empty((R, n), dtype=torch.float32, device=dev) for n in ns]
            extra["sp"] = [torch.empty((R, n + 1), dtype=torch.float32, device=dev) for n in ns]
            extra["eu"] = [torch.empty((R, n + 1), dtype=torch.float32, device=dev) for n in ns]
            for i in range(3):
                outs.weights[i] = extra["w"][i].data_ptr()
                outs.spacing_bins[i] = extra["sp"][i].data_ptr()
                outs.eucl_bins[i] = extra["eu"][i].data_ptr()

        need = lib.tn_render_workspace_bytes(rc, R)
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        _hip.check(
            lib.tn_render_rays_fwd(prop0, prop1, fld, rc, ins, outs, R, self._workspace.data_ptr(),
                                   self._workspace.numel(), _hip.current_stream()),
            "tn_render_rays_fwd",
        )
        outputs: Dict[str, Tensor] = {
            "rgb": rgb,
            RenderedImageModality.ACCUMULATION.value: buf[0][:, None],
            RenderedImageModality.DEPTH.value: buf[1][:, None],
            "expected_depth": buf[2][:, None],
        }
        if want_samples:
            outputs["weights_list"] = [w[..., None] for w in extra["w"]]
            outputs["ray_samples_list"] = [_samples_from_bins(ray_bundle, sp, eu, bool(rc.initial_sampler))
                                           for sp, eu in zip(extra["sp"], extra["eu"])]
        outputs["prop_depth_0"] = buf[3][:, None]
        outputs["prop_depth_1"] = buf[4][:, None]
        outputs[RenderedImageModality.THERMAL.value] = buf[5][:, None]
        return outputs
