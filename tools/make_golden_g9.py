"""G9  predict_normals.json — what ``predict_normals=True`` does IN THE REFERENCE, recorded by executing the reference's own code
(run ONLY in the build container, where /root/reference exists).

The reference's model forwards ``config.predict_normals`` to its field (``use_pred_normals``, thermal_nerf_model.py:108), asks the
field for normals (``compute_normals=``, :226) and then renders ``field_outputs[FieldHeadNames.PRED_NORMALS]`` (:256-258).  But
``ThermalNerfactoTField.get_outputs`` (thermal_field.py:108-181) OVERRIDES nerfstudio's ``NerfactoField.get_outputs`` and never
evaluates the predicted-normals head: the key does not exist, so every forward with ``predict_normals=True`` ends in a KeyError at
thermal_nerf_model.py:257.  The switch is part of the reference's config surface, not of its working behaviour; the product mirrors
the error (same exception type, same key) instead of inventing outputs the reference cannot produce.

Recorded, from the REAL ``ThermalNerfactoTField.forward(compute_normals=True)`` (nerfstudio's bases replaced by the G6 stand-ins
built from the oracle's primitives; ``Field.get_normals`` a shape-only stand-in) and the REAL ``ThermalNerfModel.get_outputs`` bound
to a host whose field IS that field:
  field_output_keys   the head names the reference's field returns with compute_normals=True
  exception           type name and the missing key (name and value) the model's get_outputs raises; the outputs it had assembled
                      up to that point (their keys, in order)
"""
from __future__ import annotations

import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import make_golden as G  # noqa: E402  (the stub loaders of G1-G7)


def main() -> None:
    from oracle import hotpath as H

    G._stub_modules()
    rec = G._stub_nerfstudio_field_bases(H)
    FH = rec["FieldHeadNames"]
    nerfacto = sys.modules["nerfstudio.fields.nerfacto_field"].NerfactoField

    def get_normals(self):  # NS Field.get_normals: -normalize(d density / d position); the arithmetic is not what G9 records
        return torch.zeros((*self._g9_shape, 3))

    nerfacto.get_normals = get_normals
    head = G._load(f"{G.REF}/thermo_nerf/thermal_nerf/thermal_field_head.py", "thermo_nerf.thermal_nerf.thermal_field_head")
    for name in ("thermo_nerf", "thermo_nerf.thermal_nerf", "thermo_nerf.nerfacto_config"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["thermo_nerf.thermal_nerf.thermal_field_head"] = head
    tf = G._load(f"{G.REF}/thermo_nerf/thermal_nerf/thermal_field.py", "thermo_nerf.thermal_nerf.thermal_field")
    sys.modules["thermo_nerf.thermal_nerf.thermal_field"] = tf

    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    torch.manual_seed(3)
    field = tf.ThermalNerfactoTField(aabb, hidden_dim=64, num_levels=4, max_res=64, base_res=16, features_per_level=2,
                                     log2_hashmap_size=8, hidden_dim_color=64, hidden_dim_transient=64,
                                     spatial_distortion=torch.nn.Identity(), num_images=3, use_pred_normals=True,
                                     use_average_appearance_embedding=True, appearance_embedding_dim=32, implementation="torch",
                                     use_transient_embedding=False, pass_thermal_gradients=True)
    field.eval()
    R, S = 5, 6
    pos = torch.rand(R, S, 3) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(R, 1, 3), dim=-1).expand(R, S, 3).contiguous()
    field._g9_shape = (R, S)

    class Fr:
        directions = dirs

        @staticmethod
        def get_positions():
            return pos

    class RS:
        frustums = Fr
        camera_indices = torch.zeros(R, S, 1, dtype=torch.long)
        deltas = torch.full((R, S, 1), 0.1)

        @staticmethod
        def get_weights(density):  # NS RaySamples.get_weights
            return H.get_weights(RS.deltas, density)

    out_field = field.forward(RS, compute_normals=True)
    field_keys = [getattr(k, "name", str(k)) for k in out_field.keys()]

    # ---- the model's get_outputs on that field -----------------------------------------------------------------------------------
    nn = torch.nn

    class _Any(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    import dataclasses

    mod("nerfstudio.cameras.camera_optimizers", CameraOptimizer=_Any)
    mod("nerfstudio.cameras.rays", RayBundle=object, RaySamples=object)
    mod("nerfstudio.data")
    mod("nerfstudio.data.scene_box", SceneBox=object)
    mod("nerfstudio.field_components.spatial_distortions", SpatialDistortion=nn.Module, SceneContraction=_Any)
    mod("nerfstudio.fields.density_fields", HashMLPDensityField=_Any)
    mod("nerfstudio.model_components")
    mod("nerfstudio.model_components.losses", MSELoss=nn.MSELoss, interlevel_loss=None, scale_gradients_by_distance_squared=None)
    mod("nerfstudio.model_components.ray_samplers", ProposalNetworkSampler=_Any, UniformSampler=_Any)
    mod("nerfstudio.model_components.renderers", AccumulationRenderer=_Any, DepthRenderer=_Any, NormalsRenderer=_Any, RGBRenderer=_Any)
    mod("nerfstudio.model_components.scene_colliders", NearFarCollider=_Any)
    mod("nerfstudio.model_components.shaders", NormalsShader=_Any)
    mod("nerfstudio.utils.colormaps")
    sys.modules["nerfstudio.utils"].colormaps = sys.modules["nerfstudio.utils.colormaps"]
    mod("torchmetrics")
    mod("torchmetrics.functional", structural_similarity_index_measure=None)
    mod("torchmetrics.image", PeakSignalNoiseRatio=_Any)
    mod("torchmetrics.image.lpip", LearnedPerceptualImagePatchSimilarity=_Any)

    @dataclasses.dataclass
    class ThermalNerfactoModelConfig:
        max_temperature: float = 1.0
        min_temperature: float = 0.0

    class ThermalNerfactoModel(nn.Module):
        pass

    mod("thermo_nerf.nerfacto_config.thermal_nerfacto", ThermalNerfactoModel=ThermalNerfactoModel,
        ThermalNerfactoModelConfig=ThermalNerfactoModelConfig)
    sys.modules["thermo_nerf.rendered_image_modalities"] = G._load(f"{G.REF}/thermo_nerf/rendered_image_modalities.py",
                                                                   "thermo_nerf.rendered_image_modalities")
    sys.modules["thermo_nerf.thermal_nerf.thermal_metrics"] = G._load(f"{G.REF}/thermo_nerf/thermal_nerf/thermal_metrics.py",
                                                                      "thermo_nerf.thermal_nerf.thermal_metrics")
    tr = G._load(f"{G.REF}/thermo_nerf/thermal_nerf/thermal_renderer.py", "thermo_nerf.thermal_nerf.thermal_renderer")
    sys.modules["thermo_nerf.thermal_nerf.thermal_renderer"] = tr
    ref = G._load(f"{G.REF}/thermo_nerf/thermal_nerf/thermal_nerf_model.py", "ref_thermal_nerf_model")

    host = ref.ThermalNerfModel.__new__(ref.ThermalNerfModel)
    nn.Module.__init__(host)
    host.eval()
    host.config = types.SimpleNamespace(predict_normals=True, use_gradient_scaling=False, num_proposal_iterations=2)
    host.density_fns = []
    host.field = field  # the reference's own field
    w_prop = [torch.full((R, 4, 1), 0.25), torch.full((R, 3, 1), 1.0 / 3)]
    host.proposal_sampler = lambda rb, density_fns=None: (RS, list(w_prop), [RS, RS])
    assembled = []

    def renderer(name, value):
        def call(*a, **k):
            assembled.append(name)
            return value
        return call

    host.renderer_rgb = renderer("renderer_rgb", torch.zeros(R, 3))
    host.renderer_depth = renderer("renderer_depth", torch.zeros(R, 1))
    host.renderer_expected_depth = renderer("renderer_expected_depth", torch.zeros(R, 1))
    host.renderer_accumulation = renderer("renderer_accumulation", torch.zeros(R, 1))
    host.renderer_normals = renderer("renderer_normals", torch.zeros(R, 3))
    host.normals_shader = renderer("normals_shader", torch.zeros(R, 3))
    host.thermal_renderer = tr.ThermalRenderer()
    exc = None
    try:
        ref.ThermalNerfModel.get_outputs(host, types.SimpleNamespace())
    except Exception as e:  # noqa: BLE001 - the point is to record which
        exc = e
    assert exc is not None, "the reference's get_outputs returned with predict_normals=True"
    key = exc.args[0] if exc.args else None
    doc = {
        "what": "ThermalNerfModel.get_outputs with config.predict_normals=True, executed from /root/reference (G9)",
        "field_output_keys": field_keys,
        "exception": {"type": type(exc).__name__, "key_name": getattr(key, "name", None), "key_value": getattr(key, "value", None),
                      "raised_after": assembled},
        "reference_lines": "thermal_nerf_model.py:226 (compute_normals), :252-260 (renders NORMALS, then PRED_NORMALS); "
                           "thermal_field.py:108-181 (get_outputs: no predicted-normals head), :195-200 (NORMALS only)",
    }
    out = os.path.join(G.OUT, "predict_normals.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))

    # ---- G10: use_transient_embedding=True [REF thermal_nerf_model.py:111; thermal_field.py:139-158] ----------------------------------
    # The reference's field evaluates nerfstudio's transient branch in TRAINING (embedding_transient[camera] ++ geo -> mlp_transient ->
    # two heads) and returns TRANSIENT_RGB / TRANSIENT_DENSITY; the reference's MODEL never reads either key.  Executed here: the real
    # field's output keys with the flag on (train / eval), and the real model's get_outputs on the SAME weights with the flag on and
    # off — identical keys, identical values: the flag adds parameters and wasted work, nothing else.
    MLP = sys.modules["nerfstudio.field_components.mlp"].MLP
    host.train()
    host.config.predict_normals = False
    host.camera_optimizer = types.SimpleNamespace(apply_to_raybundle=lambda rb: None)
    runs = {}
    keys = {}
    for flag in (False, True):
        torch.manual_seed(3)
        fld = tf.ThermalNerfactoTField(aabb, hidden_dim=64, num_levels=4, max_res=64, base_res=16, features_per_level=2,
                                       log2_hashmap_size=8, hidden_dim_color=64, hidden_dim_transient=64,
                                       spatial_distortion=torch.nn.Identity(), num_images=3, use_pred_normals=False,
                                       use_average_appearance_embedding=True, appearance_embedding_dim=32, implementation="torch",
                                       use_transient_embedding=flag, pass_thermal_gradients=True)
        if flag:  # what NS NerfactoField builds for the flag [NS-recall]; the G6 stand-in base does not
            torch.manual_seed(4)
            fld.transient_embedding_dim = 16
            fld.embedding_transient = torch.nn.Embedding(3, 16)
            fld.mlp_transient = MLP(15 + 16, 2, 64, 64, activation=torch.nn.ReLU())
            fld.field_head_transient_rgb = torch.nn.Sequential(torch.nn.Linear(64, 3), torch.nn.Sigmoid())
            fld.field_head_transient_density = torch.nn.Sequential(torch.nn.Linear(64, 1), torch.nn.Softplus())
        fld._g9_shape = (R, S)
        for mode in ("train", "eval"):
            fld.train(mode == "train")
            with torch.no_grad():
                o = fld.forward(RS, compute_normals=False)
            keys["flag_%s_%s" % (int(flag), mode)] = [getattr(k, "name", str(k)) for k in o.keys()]
        fld.train(True)
        host.field = fld
        calls = {}

        def rend(name):
            def call(*a, **k):
                t = [x for x in list(a) + list(k.values()) if isinstance(x, torch.Tensor)]
                calls[name] = float(sum(x.double().sum() for x in t))
                return torch.full((R, 3 if name == "renderer_rgb" else 1), calls[name] % 1.0)
            return call

        for name in ("renderer_rgb", "renderer_depth", "renderer_expected_depth", "renderer_accumulation"):
            setattr(host, name, rend(name))
        host.thermal_renderer.train(True)
        with torch.no_grad():
            out_m = ref.ThermalNerfModel.get_outputs(host, types.SimpleNamespace())
        runs[flag] = (list(out_m.keys()), {k: v.double().sum().item() for k, v in out_m.items() if isinstance(v, torch.Tensor)}, dict(calls))
    same_keys = runs[False][0] == runs[True][0]
    same_vals = all(abs(runs[False][1][k] - runs[True][1][k]) <= 1e-12 for k in runs[False][1]) and runs[False][2] == runs[True][2]
    doc2 = {"what": "use_transient_embedding=True in the reference, executed from /root/reference (G10)",
            "field_output_keys": keys, "model_output_keys": runs[True][0], "model_outputs_identical_with_and_without_the_flag": bool(same_keys and same_vals),
            "reference_lines": "thermal_field.py:139-158 (transient branch, training only); thermal_nerf_model.py:210-275 reads neither "
                               "TRANSIENT_RGB nor TRANSIENT_DENSITY"}
    with open(os.path.join(G.OUT, "transient_embedding.json"), "w") as f:
        json.dump(doc2, f, indent=1)
    print(json.dumps(doc2, indent=1))


if __name__ == "__main__":
    main()
