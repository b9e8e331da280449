"""Config surface + base class shared by the thermal models.

Mirror of [REF thermo_nerf/nerfacto_config/thermal_nerfacto.py:13-45] on top of the nerfstudio
``NerfactoModelConfig`` fields the reference consumes at [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:91-184]
(defaults = nerfstudio 1.1.5).  One value is added to ``implementation``: ``"hip"``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Literal, Tuple, Type, Union

import torch
from torch import Tensor, nn

from ..camera_optimizer import CameraOptimizerConfig
from ..rays import RayBundle
from ..scene import SceneBox


@dataclass
class NerfactoModelConfig:
    """The NerfactoModelConfig (nerfstudio 1.1.5) fields read on the ThermoNeRF path."""

    _target: Type = field(default_factory=lambda: ThermalNerfactoModel)
    near_plane: float = 0.05
    far_plane: float = 1000.0
    background_color: Literal["random", "last_sample", "black", "white"] = "last_sample"
    hidden_dim: int = 64
    hidden_dim_color: int = 64
    hidden_dim_transient: int = 64
    num_levels: int = 16
    base_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_net_args_list: List[Dict] = field(
        default_factory=lambda: [
            {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
            {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "use_linear": False},
        ]
    )
    proposal_initial_sampler: Literal["piecewise", "uniform"] = "piecewise"
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    orientation_loss_mult: float = 0.0001
    pred_normal_loss_mult: float = 0.001
    use_proposal_weight_anneal: bool = True
    use_appearance_embedding: bool = True
    use_average_appearance_embedding: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    predict_normals: bool = False
    disable_scene_contraction: bool = False
    use_gradient_scaling: bool = False
    implementation: Literal["hip", "tcnn", "torch"] = "hip"
    appearance_embed_dim: int = 32
    average_init_density: float = 0.01  # ignored by the reference (REF thermal_field.py:86 hard-codes 1.0)
    camera_optimizer: CameraOptimizerConfig = field(default_factory=lambda: CameraOptimizerConfig(mode="SO3xR3"))
    eval_num_rays_per_chunk: int = 4096
    # --- additions of this implementation -----------------------------------------------------------
    fused: bool = True
    """Run get_outputs through the single fused entry point tn_render_rays_fwd (False: one HIP call per
    nerfstudio module, mirroring the reference's call sequence)."""
    sh_input: Literal["shifted", "unit"] = "shifted"
    """Direction convention fed to the SH basis in the torch fallback (SURVEY A.6 [UNSURE])."""
    sh_direction_gradient: bool = False
    """Training: let d loss / d directions flow through the SH basis.  False (default) = nerfstudio's torch fallback, whose
    SHEncoding.pytorch_fwd runs under @torch.no_grad() (SURVEY A.6): camera-pose gradients then come from the sample
    positions only.  True = the behaviour of a differentiable (tcnn) SH encoding."""
    dense_grid_budget_mb: int = 64
    """>0: the proposal networks' eval kernels read their leading hash levels from a dense re-layout built within this budget
    per network (dense[x][y][z] = (table[hash(x,y,z)], table[hash(x,y,z+1)]): layout only, bit-identical values; the two
    z-corners of a cell are one aligned 16-byte load, so a level costs 4 gather instructions instead of 8).  64 MB holds 5
    and 4 of the reference's 2 x 5 proposal levels (45 + 41 MB); rebuilt when the tables change; never used in training."""
    field_dense_grid_budget_mb: int = 16
    """The same for the main field's 16-level grid: 16 MB holds the 6 levels (14.5 MB) the lane = ray field kernels read
    densely; a smaller budget leaves those kernels on the hashed tables."""
    use_mfma: bool = True
    """Use the MFMA form of the main-field kernel when the library provides it."""
    early_termination_eps: float = 0.0
    """eval only; > 0: a wave of 64 rays stops marching once every ray's transmittance is below this value
    (outputs move by <= eps; 0 = off = the reference's behaviour).  Only the lane = ray kernels (calls of ~60 k rays and
    more) implement it; smaller calls run one ray per wave and render exactly."""
    fused_train_forward: bool = True
    """Training: the final level's field forward (encode + mlp_base + heads, with its tape) as ONE MFMA kernel
    (tn_field_fwd_taped); False: the stage-by-stage entry points (one launch per nerfstudio module)."""
    fused_train_backward: bool = True
    """Training: the backward of each MLP (mlp_head, mlp_thermal + head, mlp_base, the proposal MLPs) as ONE launch per MLP
    (tn_linear_chain_bwd: one tile read per layer); False: one tn_linear_bwd launch per layer."""
    tape_free_training: bool = True
    """Training: the final level's field without an activation tape.  The forward (tn_field_fwd_train) keeps 38 floats per
    sample (hash features, selector, density, rgb, thermal) instead of ~380; the backward (tn_field_bwd_fused) recomputes the
    five hidden layers from the hash features in registers and runs every adjoint next to them (DESIGN §5.6).  False: the
    taped forward + chained / per-layer backward above, kept as the cross-check (same gradients, tests/test_gpu_training.py)."""
    fused_proposal_training: bool = True
    """Training, steps on which the proposal networks take gradient: each level's HashMLPDensityField forward and backward as one
    launch each (tn_density_fwd_train / tn_density_bwd_train: lane = sample, weights as scalars, hidden layer recomputed);
    False: the stage entry points + tn_linear_chain_bwd."""
    fused_backward_split: bool = True
    """tn_field_bwd_fused as three launches (colour head | thermal head | mlp_base) of two waves per SIMD each instead of one
    launch of one wave per SIMD holding all ~210 gradient accumulators (DESIGN §5.6)."""
    trunc_exp_clamp_min: float = -15.0
    """Lower clamp of trunc_exp's backward, g * exp(clamp(x, min, 15)): -15 = nerfstudio's activations.trunc_exp (taken from
    torch-ngp, two-sided); float("-inf") = upper clamp only (SURVEY A.3 [UNSURE])."""
    store_base_output: bool = True
    """Training (round 5, with tape_free_training and fused_backward_split): the forward also keeps mlp_base's 16 output rows
    (64 B per sample, 54 floats per sample in all) and the backward's colour and thermal launches read them instead of
    recomputing mlp_base from the hash features — 48 of a head tile's ~330 MFMAs (DESIGN §5.6)."""
    store_position_jacobian: bool = True
    """Training with ray gradients (camera-pose optimisation; round 5, with tape_free_training and fused_backward_split): the forward
    also writes d hash features / d position (96 floats per sample) while the eight corner values of every level are in its
    registers, and the backward's position gradient is 12 streamed reads per lane instead of 32 table reads (DESIGN §5.6)."""
    backward_bf16_pieces: bool = True
    """Training (round 5, with store_base_output): the 64 x 64 products of the backward's two head launches — the second layer's
    recomputed forward and its dx — on the bf16 matrix cores as six-product splits of three exact bf16 pieces per operand
    (fp32's rounding size per product, the arithmetic of mlp_precision="bf16x6"); the weight-gradient products stay on the fp32
    MFMA (DESIGN §5.6).  False: every product on v_mfma_f32_16x16x4_f32."""
    bucketed_table_scatter: bool = True
    """Training: hash-table gradient of the field's fine levels (scaling >= 200: levels 8-15) as bucketed records + LDS sums instead of
    global atomics (tn_hash_encode_bwd_sorted), the coarse levels with the atomics: 5.03 against 5.25 ms per step at S=192,
    2.24 against 2.27 at S=48 (DESIGN §5.6)."""
    spread_coarse_scatter: bool = True
    """Training: the coarsest levels of the table-gradient scatter (dense vertex grid <= 48^3: the first four levels of the
    reference grids) accumulate into 16 private dense copies that a second launch sums and hashes into the gradient
    (tn_hash_encode_bwd_spread): same-address atomics retire one at a time, and at those levels a trained scene's samples
    share a few thousand entries (DESIGN §5.6)."""
    overlap_table_scatter: bool = True
    """Training: the bucketed part of the field's table-gradient scatter runs on a second HIP stream beside the atomic part —
    disjoint levels of the gradient, one waiting on the memory-side atomic unit, the other on LDS and streaming (DESIGN §5.6)."""
    fused_step_calls: bool = True
    """Training (round 6): on the steps whose proposal networks take no gradient (5 of 6 after warm-up) the forward's launch chain and
    the backward's are ONE C-ABI call each (tn_train_step_fwd / tn_train_step_bwd: the same entry points, order and streams as the
    per-call host path, issued from C++; the step's per-sample tensors in one slab) — ~20 ctypes calls and ~45 allocations per step
    less on the host, which decides the step at the reference's default S = 48.  False: one call per launch (the cross-check)."""
    deferred_table_update: bool = False
    """Training (round 6; set by thermo_nerf_amd.trainer.Trainer with its HipAdam optimizer, off for a foreign training loop): the field's
    table-gradient scatter — both halves — and the table's Adam run on the step's side streams and are NOT joined by the backward:
    the calling stream goes on with everything that needs neither the table nor its gradient (ray-level adjoints, camera-pose
    backward, the small tensors' Adam, the next step's ray gather / camera optimizer / proposal pass / field_prepare) and joins right
    before the next field forward.  Whoever reads ``hash_table`` or its ``.grad`` on another stream in between calls
    ``thermo_nerf_amd._hip.join_pending()`` first (Module.train / state_dict / the eval paths / Trainer.train's exit do)."""
    overlap_regularisers: Union[bool, str] = "auto"
    """Training: the distortion and interlevel terms are launched by the forward itself, on the step's side streams beside
    the depth renderers and the image losses (they depend on the forward's weights and bins only); get_metrics_dict /
    get_loss_dict pick the results up.  False: each is launched where it is asked for; "auto": on from 4096 x 96 final-level
    samples per step, where the step's device time leaves room for the extra stream joins on the host (DESIGN §5.6)."""
    kernel_family: Literal["auto", "lane_ray", "ray_per_wave"] = "auto"
    """Which form of the fused kernels a call runs (tn_render_config.kernel_family): "auto" picks by call size (lane = ray —
    one wave owns 64 consecutive rays — from ~60-80 k rays up, one ray per wave below); the other two force a form."""
    sample_split: int = 0
    """Sample-split tiles of the exact-fp32 lane = ray field kernel (tn_render_config.sample_split): 0 = the library picks the
    number of segments a 64-ray tile's sample march is cut into from the call's size (1 from ~400 k rays up; a 65 536-ray chunk: 2),
    1 = never, k = k segments.  Eval only; same tolerances, other last bits than the serial march (DESIGN §7)."""
    mlp_precision: Literal["f32", "bf16x6", "f16x3"] = "f32"
    """"f32": exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  Eval-only alternatives on the matrix cores' 16-bit rate, fp32 accumulate:
    "bf16x6" — every fp32 operand as three bf16 pieces (24 bits: an exact split), six piece products per fp32 product, 2^-23
    relative per-product error = the size of fp32's own rounding (an fp32 dot product in another order); "f16x3" — two f16
    pieces (22 bits), three products, ~2^-22 (activations must stay below 65504)."""

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


@dataclass
class ThermalNerfactoModelConfig(NerfactoModelConfig):
    """[REF thermal_nerfacto.py:13-25]"""

    _target: Type = field(default_factory=lambda: ThermalNerfactoModel)
    max_temperature: float = 1.0
    min_temperature: float = 0.0
    cold: bool = False
    camera_optimizer_mode: Literal["off", "SO3xR3", "SE3"] = "SO3xR3"


KERNEL_FAMILY = {"auto": 0, "lane_ray": 1, "ray_per_wave": 2}


class ThermalNerfactoModel(nn.Module):
    """Base of the thermal models [REF thermal_nerfacto.py:28-45] with the slice of nerfstudio's ``Model`` the hot
    path needs: ``forward`` (collider -> get_outputs) and chunked ``get_outputs_for_camera_ray_bundle``."""

    config: ThermalNerfactoModelConfig

    def __init__(self, config: ThermalNerfactoModelConfig, scene_box: SceneBox, num_train_data: int, **kwargs) -> None:
        super().__init__()
        config.camera_optimizer = CameraOptimizerConfig(mode=config.camera_optimizer_mode)  # REF :38-40
        self.config = config
        self.scene_box = scene_box
        self.num_train_data = num_train_data
        self.kwargs = kwargs
        self.collider = None
        self.max_temperature = config.max_temperature
        self.min_temperature = config.min_temperature
        self.populate_modules()
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self) -> None:  # pragma: no cover - overridden
        raise NotImplementedError

    def get_outputs(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:  # pragma: no cover - overridden
        raise NotImplementedError

    def forward(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """NS Model.forward: collider, then get_outputs."""
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        return self.get_outputs(ray_bundle)

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """NS Model.get_outputs_for_camera_ray_bundle: [H,W] bundle, row-major chunks."""
        num_rays_per_chunk = self.config.eval_num_rays_per_chunk
        image_height, image_width = camera_ray_bundle.origins.shape[:2]
        num_rays = len(camera_ray_bundle)
        outputs_lists: Dict[str, List[Tensor]] = {}
        for i in range(0, num_rays, num_rays_per_chunk):
            ray_bundle = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + num_rays_per_chunk).to(self.device)
            outputs = self.forward(ray_bundle=ray_bundle)
            for name, out in outputs.items():
                if not isinstance(out, Tensor):
                    continue
                outputs_lists.setdefault(name, []).append(out)
        return {k: torch.cat(v).view(image_height, image_width, -1) for k, v in outputs_lists.items()}
