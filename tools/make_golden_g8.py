"""G8  populate_modules.json — run ONLY in the build container (needs /root/reference).

Executes the REAL ``ThermalNerfModel.__init__`` / ``populate_modules`` [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:67-208]
and the real ``ThermalNerfactoModel.__init__`` [REF thermo_nerf/nerfacto_config/thermal_nerfacto.py:31-45] with every nerfstudio
constructor they call (and ``ThermalNerfactoTField`` / ``ThermalRenderer``) replaced by the recording stand-ins of
tests/g8_harness.py, for the configurations of ``g8_harness.CONFIG_VARIANTS``; stores, per configuration, every constructor call
in order with its keyword values, which attribute holds which module, ``density_fns`` and ``update_schedule(step)`` at a few
steps.  Then executes the REAL method config [REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:17-48] with nerfstudio's config
classes replaced by kwargs recorders and stores the values it sets.  The fixture is data: names and numbers, no source text.
"""
from __future__ import annotations

import dataclasses
import importlib.util
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "populate_modules.json")

from tests import g8_harness as G  # noqa: E402


def _load(path: str, name: str):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def mod(name: str, **attrs):
    m = sys.modules.get(name) or types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(mod(parent), leaf, m)
    return m


class _Sub:
    def __class_getitem__(cls, item):
        return cls


def reference_populate_modules(variant: dict) -> dict:
    log = G.Log()
    R = lambda n: G.recorder(n, log)  # noqa: E731
    for n in [k for k in sys.modules if k.startswith(("nerfstudio", "thermo_nerf", "torchmetrics", "jaxtyping", "ref_"))]:
        del sys.modules[n]
    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Int", "Shaped"):
        setattr(jt, n, type(n, (_Sub,), {}))
    sys.modules["jaxtyping"] = jt

    ns_fields = [k for k in variant if k not in ("max_temperature", "min_temperature", "cold", "camera_optimizer_mode",
                                                "use_transient_embedding", "pass_thermal_gradients")]
    NerfactoModelConfig = dataclasses.make_dataclass(  # stands for nerfstudio's: only the field names matter, values come per variant
        "NerfactoModelConfig", [(k, object, dataclasses.field(default=None)) for k in ns_fields])

    class NerfactoModel(torch.nn.Module):  # NS Model.__init__: keeps its arguments and calls populate_modules()
        def __init__(self, config, scene_box, num_train_data, **kwargs):
            super().__init__()
            self.config, self.scene_box, self.num_train_data, self.kwargs = config, scene_box, num_train_data, kwargs
            self.populate_modules()

    mod("nerfstudio.cameras.camera_optimizers", CameraOptimizer=object, CameraOptimizerConfig=R("CameraOptimizerConfig"))
    mod("nerfstudio.cameras.rays", RayBundle=object, RaySamples=object)
    mod("nerfstudio.data.scene_box", SceneBox=object)
    mod("nerfstudio.models.nerfacto", NerfactoModel=NerfactoModel, NerfactoModelConfig=NerfactoModelConfig)
    mod("nerfstudio.field_components.field_heads", FieldHeadNames=object)
    mod("nerfstudio.field_components.spatial_distortions", SceneContraction=R("SceneContraction"))
    mod("nerfstudio.fields.density_fields", HashMLPDensityField=R("HashMLPDensityField"))
    mod("nerfstudio.model_components.losses", MSELoss=R("MSELoss"), interlevel_loss=None, scale_gradients_by_distance_squared=None)
    mod("nerfstudio.model_components.ray_samplers", ProposalNetworkSampler=R("ProposalNetworkSampler"), UniformSampler=R("UniformSampler"))
    mod("nerfstudio.model_components.renderers", AccumulationRenderer=R("AccumulationRenderer"), DepthRenderer=R("DepthRenderer"),
        NormalsRenderer=R("NormalsRenderer"), RGBRenderer=R("RGBRenderer"))
    mod("nerfstudio.model_components.scene_colliders", NearFarCollider=R("NearFarCollider"))
    mod("nerfstudio.model_components.shaders", NormalsShader=R("NormalsShader"))
    mod("nerfstudio.utils.colormaps")
    mod("torchmetrics.functional", structural_similarity_index_measure=None)
    mod("torchmetrics.image", PeakSignalNoiseRatio=R("PeakSignalNoiseRatio"))
    mod("torchmetrics.image.lpip", LearnedPerceptualImagePatchSimilarity=R("LearnedPerceptualImagePatchSimilarity"))
    for name in ("thermo_nerf", "thermo_nerf.thermal_nerf", "thermo_nerf.nerfacto_config"):
        mod(name)
    _load(f"{REF}/thermo_nerf/rendered_image_modalities.py", "thermo_nerf.rendered_image_modalities")
    _load(f"{REF}/thermo_nerf/thermal_nerf/thermal_metrics.py", "thermo_nerf.thermal_nerf.thermal_metrics")
    _load(f"{REF}/thermo_nerf/nerfacto_config/thermal_nerfacto.py", "thermo_nerf.nerfacto_config.thermal_nerfacto")  # the REAL base
    mod("thermo_nerf.thermal_nerf.thermal_field", ThermalNerfactoTField=R("ThermalNerfactoTField"))
    mod("thermo_nerf.thermal_nerf.thermal_field_head", FieldHeadNamesT=object)
    mod("thermo_nerf.thermal_nerf.thermal_renderer", ThermalRenderer=R("ThermalRenderer"))
    ref = _load(f"{REF}/thermo_nerf/thermal_nerf/thermal_nerf_model.py", "ref_thermal_nerf_model")

    cfg = ref.ThermalNerfModelConfig(**variant)
    box = types.SimpleNamespace(aabb=torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]))
    try:
        ref.ThermalNerfModel(cfg, metadata={}, scene_box=box, num_train_data=G.NUM_TRAIN_DATA)
        raise AssertionError("missing 'thermal' metadata must raise")
    except ValueError as e:
        missing = str(e)
    log.calls.clear()
    log.instances.clear()
    model = ref.ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=box, num_train_data=G.NUM_TRAIN_DATA)
    out = G.summarize(model, log)
    out["missing_thermal_error"] = missing
    out["model_attributes"] = {"max_temperature": model.max_temperature, "min_temperature": model.min_temperature, "step": model.step}
    return out


def reference_method_config() -> dict:
    """config_thermal_nerf.py executed with kwargs recorders in place of nerfstudio's config classes"""
    class KW:
        def __init__(self, **kw):
            self.kw = kw

        def __class_getitem__(cls, item):
            return cls

    names = {}
    for n in ("ViewerConfig", "VanillaDataManager", "VanillaDataManagerConfig", "AdamOptimizerConfig",
              "ExponentialDecaySchedulerConfig", "TrainerConfig", "VanillaPipelineTrackingConfig", "ThermalDataParserConfig",
              "ThermalDataset", "ThermalNerfModelConfig"):
        names[n] = type(n, (KW,), {})
    for n in [k for k in sys.modules if k.startswith(("nerfstudio", "thermo_nerf"))]:
        del sys.modules[n]
    mod("nerfstudio.configs.base_config", ViewerConfig=names["ViewerConfig"])
    mod("nerfstudio.data.datamanagers.base_datamanager", VanillaDataManager=names["VanillaDataManager"],
        VanillaDataManagerConfig=names["VanillaDataManagerConfig"])
    mod("nerfstudio.engine.optimizers", AdamOptimizerConfig=names["AdamOptimizerConfig"])
    mod("nerfstudio.engine.schedulers", ExponentialDecaySchedulerConfig=names["ExponentialDecaySchedulerConfig"])
    mod("nerfstudio.engine.trainer", TrainerConfig=names["TrainerConfig"])
    mod("thermo_nerf.nerfstudio_config.pipeline_tracking", VanillaPipelineTrackingConfig=names["VanillaPipelineTrackingConfig"])
    mod("thermo_nerf.thermal_nerf.thermal_dataparser", ThermalDataParserConfig=names["ThermalDataParserConfig"])
    mod("thermo_nerf.thermal_nerf.thermal_dataset", ThermalDataset=names["ThermalDataset"])
    mod("thermo_nerf.thermal_nerf.thermal_nerf_model", ThermalNerfModelConfig=names["ThermalNerfModelConfig"])
    m = _load(f"{REF}/thermo_nerf/thermal_nerf/config_thermal_nerf.py", "ref_config_thermal_nerf")

    def enc(v):
        if isinstance(v, KW):
            return {"__config__": type(v).__name__, **{k: enc(x) for k, x in v.kw.items()}}
        if isinstance(v, dict):
            return {k: enc(x) for k, x in v.items()}
        if isinstance(v, type):
            return {"__class__": v.__name__}
        return v

    return enc(m.thermal_nerf_config)


def main() -> None:
    assert os.path.isdir(REF), "the reference checkout is needed to generate G8"
    out = {"populate_modules": {name: reference_populate_modules(v) for name, v in G.CONFIG_VARIANTS.items()},
           "method_config": reference_method_config()}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
