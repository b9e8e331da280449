"""ThermalNerfactoTField — the nerfacto field plus ThermoNeRF's thermal branch, evaluated on MI355X.

Interface mirror of [REF thermo_nerf/thermal_nerf/thermal_field.py:33-201] (constructor arguments, module and
parameter names, ``get_density`` / ``get_outputs`` / ``forward`` signatures, returned dict keys, error
behaviour).  The arithmetic is two HIP kernels: ``tn_field_density_fwd`` (NS NerfactoField.get_density) and
``tn_field_heads_fwd`` (REF :108-181: SH + appearance + mlp_head, and mlp_thermal + ThermalFieldHead).
"""
from __future__ import annotations

from typing import Dict, Literal, Optional, Tuple, Union

import torch
from torch import Tensor, nn

from .. import _hip
from ..fields import Embedding, FieldHeadNames, MLP, MLPWithHashEncoding
from ..rays import RaySamples
from ..scene import SceneContraction
from .thermal_field_head import FieldHeadNamesT, ThermalFieldHead


class ThermalNerfactoTField(nn.Module):
    def __init__(
        self,
        aabb: Tensor,
        num_images: int,
        num_layers: int = 2,
        hidden_dim: int = 64,
        geo_feat_dim: int = 15,
        num_levels: int = 16,
        base_res: int = 16,
        max_res: int = 2048,
        log2_hashmap_size: int = 19,
        num_layers_color: int = 3,
        num_layers_transient: int = 2,
        features_per_level: int = 2,
        hidden_dim_color: int = 64,
        hidden_dim_transient: int = 64,
        appearance_embedding_dim: int = 32,
        transient_embedding_dim: int = 16,
        use_transient_embedding: bool = False,
        use_semantics: bool = False,
        num_semantic_classes: int = 100,
        pass_semantic_gradients: bool = False,
        use_pred_normals: bool = False,
        use_average_appearance_embedding: bool = False,
        spatial_distortion: Optional[SceneContraction] = None,
        implementation: Literal["hip", "tcnn", "torch"] = "hip",
        pass_thermal_gradients: bool = False,
        sh_input: Literal["shifted", "unit"] = "shifted",
    ) -> None:
        super().__init__()
        if use_semantics:
            # REF thermal_nerf_model.py:96-114 never passes it (NerfactoField's default False)
            raise NotImplementedError("the semantics head is off on the ThermoNeRF path")
        if (num_layers, num_layers_color, num_layers_transient) != (2, 3, 2):
            raise NotImplementedError("kernels implement mlp_base 2, mlp_head 3, mlp_thermal 2 layers (the defaults)")
        widths = (int(hidden_dim), int(hidden_dim_color), int(hidden_dim_transient))
        if min(widths) < 1 or max(widths) > 256:
            raise NotImplementedError("hidden_dim / hidden_dim_color / hidden_dim_transient [REF thermal_nerf_model.py:96-114] up to 256 "
                                      "(tn_linear_fwd / tn_linear_bwd's limit); the defaults are 64")
        # the fused kernels (MFMA chains, one launch per pass) are laid out for the reference's 64-wide layers; any other width runs
        # the SAME arithmetic stage by stage — hash encode, one tn_linear_fwd / tn_linear_bwd per layer — in eval and in training
        self.staged = widths != (64, 64, 64)
        self.register_buffer("aabb", aabb.clone().float())
        self.geo_feat_dim = geo_feat_dim
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.spatial_distortion = spatial_distortion
        self.num_images = num_images
        self.appearance_embedding_dim = appearance_embedding_dim
        self.use_average_appearance_embedding = use_average_appearance_embedding
        self.use_transient_embedding = use_transient_embedding
        self.use_semantics = use_semantics
        self.use_pred_normals = use_pred_normals
        self.pass_semantic_gradients = pass_semantic_gradients
        # REF thermal_field.py:86 passes average_init_density = 1.0 positionally to NerfactoField
        self.average_init_density = 1.0
        self.sh_input = sh_input

        self.embedding_appearance = Embedding(self.num_images, self.appearance_embedding_dim)
        self.mlp_base = MLPWithHashEncoding(num_levels, base_res, max_res, log2_hashmap_size, features_per_level,
                                            num_layers, hidden_dim, 1 + self.geo_feat_dim)
        sh_dim = 16  # SHEncoding(levels=4)
        self.mlp_head = MLP(sh_dim + self.geo_feat_dim + self.appearance_embedding_dim, num_layers_color,
                            hidden_dim_color, 3)
        # REF thermal_field.py:90-98: 15 -> 64 -> 64, ReLU, Sigmoid
        self.mlp_thermal = MLP(self.geo_feat_dim, 2, 64, hidden_dim_transient)
        self.field_head_thermal = ThermalFieldHead(in_dim=self.mlp_thermal.get_out_dim())  # REF :100-102
        self.transient_embedding_dim = transient_embedding_dim
        if use_transient_embedding:
            # config.use_transient_embedding=True [REF thermal_nerf_model.py:111]: NS NerfactoField builds the transient branch —
            # Embedding(num_images, 16), MLP(15 + 16 -> hidden_dim_transient, 2 layers), uncertainty / rgb / density heads
            # [NS-recall] — and the reference's field evaluates it in training [REF thermal_field.py:139-158], but the reference's
            # MODEL reads neither TRANSIENT_RGB nor TRANSIENT_DENSITY [REF thermal_nerf_model.py:210-275]: recorded by executing it
            # (G10, tests/golden/transient_embedding.json: model outputs identical with the flag on and off).  So the parameters
            # exist (module tree / state-dict names are the reference's; they never receive a gradient there either) and the
            # branch's two unread dictionary entries are not evaluated.
            self.embedding_transient = Embedding(self.num_images, transient_embedding_dim)
            self.mlp_transient = MLP(self.geo_feat_dim + transient_embedding_dim, num_layers_transient, hidden_dim_transient,
                                     hidden_dim_transient)
            for name, width in (("uncertainty", 1), ("rgb", 3), ("density", 1)):
                head = nn.Module()
                head.net = nn.Linear(self.mlp_transient.get_out_dim(), width)
                setattr(self, "field_head_transient_" + name, head)
        if use_pred_normals:
            # config.predict_normals=True [REF thermal_nerf_model.py:108]: NS NerfactoField builds the predicted-normals head —
            # NeRFEncoding(3, 2 frequencies) -> MLP(15 + 12, 3 layers of 64, out hidden_dim_transient) -> PredNormalsFieldHead
            # (Linear -> 3, Tanh) [NS-recall] — and the reference's get_outputs override never evaluates it (G9: the model ends in
            # KeyError(PRED_NORMALS)).  The parameters exist so that the module tree and state-dict names are the reference's.
            self.mlp_pred_normals = MLP(self.geo_feat_dim + 12, 3, 64, hidden_dim_transient)
            self.field_head_pred_normals = nn.Module()
            self.field_head_pred_normals.net = nn.Linear(self.mlp_pred_normals.get_out_dim(), 3)
        self.pass_thermal_gradients = pass_thermal_gradients
        self.training_iteration = 0
        self.pass_rgb_gradients = True
        self.dense_budget_bytes = 0
        self._prepared: Optional[Tensor] = None
        self._prepared_h3: Optional[Tensor] = None
        self._prepared_b6: Optional[Tensor] = None
        self._prepared_key = None

    # ------------------------------------------------------------------------------------------------
    def c_struct(self, prepare: bool = False, precision: str = "f32", dense: bool = True) -> _hip.tn_thermal_field:
        _hip.join_pending()  # (config.deferred_table_update) the struct's users read the table on the calling stream
        f = _hip.tn_thermal_field()
        f.grid = self.mlp_base.encoder.c_struct(self.dense_budget_bytes if dense else 0)
        f.base0 = _hip.make_linear(self.mlp_base.mlp.layers[0])
        f.base1 = _hip.make_linear(self.mlp_base.mlp.layers[1])
        f.head0 = _hip.make_linear(self.mlp_head.layers[0])
        f.head1 = _hip.make_linear(self.mlp_head.layers[1])
        f.head2 = _hip.make_linear(self.mlp_head.layers[2])
        f.th0 = _hip.make_linear(self.mlp_thermal.layers[0])
        f.th1 = _hip.make_linear(self.mlp_thermal.layers[1])
        f.thead = _hip.make_linear(self.field_head_thermal.net)
        emb = _hip.require_device_tensor(self.embedding_appearance.embedding.weight.detach(), "embedding_appearance")
        f.appearance = emb.data_ptr()
        f.num_images = self.num_images
        f.app_dim = self.appearance_embedding_dim
        f.geo_feat_dim = self.geo_feat_dim
        f.use_average_appearance = 1 if self.use_average_appearance_embedding else 0
        f.sh_shifted = 1 if self.sh_input == "shifted" else 0
        f.space = _hip.make_space(self.spatial_distortion is not None, self.aabb, owner=self)
        f.average_init_density = float(self.average_init_density)
        f.prepared = None
        f.prepared_f16x3 = None
        f.prepared_bf16x6 = None
        if prepare:
            lib = _hip.load()
            plist = self.__dict__.get("_tn_plist")
            if plist is None:
                plist = self.__dict__["_tn_plist"] = list(self.parameters())
            key = tuple([(p.data_ptr(), p._version) for p in plist])
            if self._prepared_key != key:
                self._prepared, self._prepared_h3, self._prepared_b6 = None, None, None
                nbytes = lib.tn_field_prepare_bytes(f)
                if nbytes > 0:
                    self._prepared = torch.empty(nbytes, dtype=torch.uint8, device=emb.device)
                    _hip.check(lib.tn_field_prepare(f, self._prepared.data_ptr(), nbytes, _hip.current_stream()),
                               "tn_field_prepare")
                self._prepared_key = key
            f.prepared = None if self._prepared is None else self._prepared.data_ptr()
            if precision == "f16x3":
                if self._prepared_h3 is None:  # built on first use for the current weights (the training path never needs it)
                    nbytes = lib.tn_field_prepare_f16x3_bytes(f)
                    if nbytes > 0:
                        self._prepared_h3 = torch.empty(nbytes, dtype=torch.uint8, device=emb.device)
                        _hip.check(lib.tn_field_prepare_f16x3(f, self._prepared_h3.data_ptr(), nbytes, _hip.current_stream()),
                                   "tn_field_prepare_f16x3")
                f.prepared_f16x3 = None if self._prepared_h3 is None else self._prepared_h3.data_ptr()
            elif precision == "bf16x6":
                if self._prepared_b6 is None:
                    nbytes = lib.tn_field_prepare_bf16x6_bytes(f)
                    if nbytes > 0:
                        self._prepared_b6 = torch.empty(nbytes, dtype=torch.uint8, device=emb.device)
                        _hip.check(lib.tn_field_prepare_bf16x6(f, self._prepared_b6.data_ptr(), nbytes, _hip.current_stream()),
                                   "tn_field_prepare_bf16x6")
                f.prepared_bf16x6 = None if self._prepared_b6 is None else self._prepared_b6.data_ptr()
            elif precision != "f32":
                raise ValueError(f"mlp_precision {precision!r}: expected 'f32', 'bf16x6' or 'f16x3'")
        return f

    def train_struct(self, prepare: bool = False) -> Optional[_hip.tn_thermal_field]:
        """The training step's ``c_struct(dense=False)``, kept for as long as the parameters keep their storage (an optimizer
        updates them in place, so the pointers stay right).  ``prepare``: the same struct with ``prepared`` pointing at ONE
        buffer per module that ``tn_field_prepare`` refills on every call (the weights have changed since the last step; calls
        on one stream are ordered) — None for a geometry the MFMA chain does not cover.  Read-only for the caller."""
        plist = self.__dict__.get("_tn_plist")
        if plist is None:
            plist = self.__dict__["_tn_plist"] = list(self.parameters())
        # (the struct holds the scene box by VALUE: an in-place change of the buffer — a checkpoint with another box — must rebuild it)
        ptrs = tuple([p.data_ptr() for p in plist]) + (self.aabb.data_ptr(), self.aabb._version, self.spatial_distortion is not None,
                                                        self.sh_input, float(self.average_init_density),
                                                        bool(self.use_average_appearance_embedding),
                                                        _hip.current_stream() if prepare else 0)
        hit = self.__dict__.get("_tn_train_struct")
        if hit is None or hit[0] != ptrs[:-1]:
            hit = self.__dict__["_tn_train_struct"] = (ptrs[:-1], self.c_struct(prepare=False, dense=False))
            self.__dict__.pop("_tn_train_prepared", None)
        raw = hit[1]
        if not prepare:
            return raw
        lib = _hip.load()
        prep = self.__dict__.get("_tn_train_prepared")
        if prep is None or prep[0] != ptrs[-1]:
            nbytes = lib.tn_field_prepare_bytes(raw)
            buf = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=plist[0].device)
            st = _hip.tn_thermal_field.from_buffer_copy(raw)
            st.prepared = buf.data_ptr() if nbytes > 0 else None
            prep = self.__dict__["_tn_train_prepared"] = (ptrs[-1], buf, st, nbytes)
        if prep[3] == 0:
            return None
        if prepare == "struct":  # (raw, the struct whose `prepared` blob the caller refills — tn_train_step_fwd does —, its size)
            return raw, prep[2], prep[3]
        _hip.check(lib.tn_field_prepare(raw, prep[1].data_ptr(), prep[3], _hip.current_stream()), "tn_field_prepare")
        return prep[2]

    # ------------------------------------------------------------------------------------------------
    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        """NS NerfactoField.get_density: (density [...,1], geo embedding [...,geo_feat_dim])."""
        positions = ray_samples.frustums.get_positions()
        return self.density_at(positions)

    def density_at(self, positions: Tensor) -> Tuple[Tensor, Tensor]:
        pos = _hip.require_device_tensor(positions, "positions")
        flat = pos.reshape(-1, 3)
        n = flat.shape[0]
        if self.staged:
            return self._density_staged(positions, flat, n)
        density = torch.empty((n,), dtype=torch.float32, device=flat.device)
        geo = torch.empty((n, self.geo_feat_dim), dtype=torch.float32, device=flat.device)
        lib = _hip.load()
        _hip.check(lib.tn_field_density_fwd(self.c_struct(), flat.data_ptr(), n, density.data_ptr(), geo.data_ptr(),
                                            _hip.current_stream()), "tn_field_density_fwd")
        shape = positions.shape[:-1]
        return density.view(*shape, 1), geo.view(*shape, self.geo_feat_dim)

    # ---- widths other than 64: one launch per nerfstudio module / layer (the training step's stage entry points) --------------
    def _density_staged(self, positions: Tensor, flat: Tensor, n: int) -> Tuple[Tensor, Tensor]:
        """NS NerfactoField.get_density as its modules: HashEncoding -> Linear + ReLU -> Linear -> split -> trunc_exp * selector"""
        from .. import training as TR

        fld = self.train_struct()
        enc, sel = TR.hash_encode_fwd(fld.grid, fld.space, flat)
        h1 = TR.linear_fwd(enc, 0, enc.shape[1], fld.base0, TR.ACT_RELU, n)
        bo = TR.linear_fwd(h1, 0, h1.shape[1], fld.base1, TR.ACT_NONE, n)  # [n, 1 + geo]
        density = torch.empty((n,), dtype=torch.float32, device=flat.device)
        _hip.check(_hip.load().tn_density_act_fwd(bo.data_ptr(), bo.shape[1], sel.data_ptr(), fld.average_init_density, n,
                                                  density.data_ptr(), _hip.current_stream()), "tn_density_act_fwd")
        shape = positions.shape[:-1]
        return density.view(*shape, 1), bo[:, 1:].contiguous().view(*shape, self.geo_feat_dim)

    def _heads_staged(self, dirs: Tensor, geo: Tensor, cam: Optional[Tensor], n: int) -> Tuple[Tensor, Tensor]:
        """[REF thermal_field.py:117-179] as its modules: [SH | geo | appearance] -> mlp_head; geo -> mlp_thermal -> thermal head"""
        from .. import training as TR

        lib = _hip.load()
        fld = self.c_struct(prepare=False, dense=False) if not self.training else self.train_struct()
        cin = torch.empty((n, 64), dtype=torch.float32, device=dirs.device)
        G = self.geo_feat_dim
        _hip.check(lib.tn_color_input_fwd(fld, dirs.data_ptr(), geo.data_ptr(), G, _hip.ptr(cam), 1 if self.training else 0, n, 1,
                                          cin.data_ptr(), _hip.current_stream()), "tn_color_input_fwd")
        c1 = TR.linear_fwd(cin, 0, 64, fld.head0, TR.ACT_RELU, n)
        c2 = TR.linear_fwd(c1, 0, c1.shape[1], fld.head1, TR.ACT_RELU, n)
        rgb = TR.linear_fwd(c2, 0, c2.shape[1], fld.head2, TR.ACT_SIGMOID, n)
        t1 = TR.linear_fwd(geo, 0, G, fld.th0, TR.ACT_RELU, n)
        t2 = TR.linear_fwd(t1, 0, t1.shape[1], fld.th1, TR.ACT_SIGMOID, n)
        thermal = TR.linear_fwd(t2, 0, t2.shape[1], fld.thead, TR.ACT_NONE, n)
        return rgb, thermal.view(n)

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None
                    ) -> Dict[Union[FieldHeadNamesT, FieldHeadNames], Tensor]:
        """REF thermal_field.py:108-181: RGB via mlp_head on [SH(dir), geo, appearance]; THERMAL via
        mlp_thermal + ThermalFieldHead on geo."""
        assert density_embedding is not None
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        dirs_src = ray_samples.frustums.directions
        outputs_shape = dirs_src.shape[:-1]
        dirs = _hip.require_device_tensor(dirs_src.reshape(-1, 3), "directions")
        geo = _hip.require_device_tensor(density_embedding.reshape(-1, self.geo_feat_dim), "density_embedding")
        n = dirs.shape[0]
        cam = None
        if self.training:
            cam = _hip.require_device_tensor(ray_samples.camera_indices.reshape(-1).to(torch.int32), "camera_indices",
                                             torch.int32)
        if self.staged:
            rgb, thermal = self._heads_staged(dirs, geo, cam, n)
            return {FieldHeadNames.RGB: rgb.view(*outputs_shape, 3), FieldHeadNamesT.THERMAL: thermal.view(*outputs_shape, 1)}
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dirs.device)
        thermal = torch.empty((n,), dtype=torch.float32, device=dirs.device)
        lib = _hip.load()
        _hip.check(
            lib.tn_field_heads_fwd(self.c_struct(), dirs.data_ptr(), geo.data_ptr(), _hip.ptr(cam), n,
                                   1 if self.training else 0, rgb.data_ptr(), thermal.data_ptr(),
                                   _hip.current_stream()),
            "tn_field_heads_fwd",
        )
        return {
            FieldHeadNames.RGB: rgb.view(*outputs_shape, 3),
            FieldHeadNamesT.THERMAL: thermal.view(*outputs_shape, 1),
        }

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False
                ) -> Dict[Union[FieldHeadNamesT, FieldHeadNames], Tensor]:
        """REF thermal_field.py:183-201."""
        if compute_normals:
            # the only caller that passes True is the model with config.predict_normals, which the reference cannot run
            # (KeyError on PRED_NORMALS in every forward: G9, tests/golden/predict_normals.json) — the analytic normals it would
            # compute on the way [REF :195-200] are never returned to anyone
            raise NotImplementedError("compute_normals=True: the reference's model raises KeyError(PRED_NORMALS) right after asking "
                                      "for them (G9); analytic normals are not evaluated on this path")
        density, density_embedding = self.get_density(ray_samples)
        field_outputs = self.get_outputs(ray_samples, density_embedding=density_embedding)
        field_outputs[FieldHeadNames.DENSITY] = density
        return field_outputs
