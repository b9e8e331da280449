"""Recording harness shared by tools/make_golden_g8.py (which runs the REFERENCE's ``populate_modules`` and method config in the
build container) and tests/test_populate_modules.py (which runs the PRODUCT's): constructor stand-ins that note, in call order,
which class was built with which keyword values, and an encoder that turns those values into JSON."""
from __future__ import annotations

from typing import Any, Dict, List

import torch
from torch import nn

# the config fields populate_modules consumes [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:91-187], each with a value
# no other field has, so that a swapped routing shows up in the record
CONFIG_VARIANTS: Dict[str, Dict[str, Any]] = {}
_BASE = dict(
    near_plane=0.07, far_plane=900.0, background_color="last_sample", hidden_dim=61, hidden_dim_color=59,
    hidden_dim_transient=53, num_levels=13, base_res=17, max_res=1999, log2_hashmap_size=18, features_per_level=3,
    num_proposal_samples_per_ray=(255, 95), num_nerf_samples_per_ray=47, proposal_update_every=5, proposal_warmup=5000,
    num_proposal_iterations=2, use_same_proposal_network=False,
    proposal_net_args_list=[{"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
                            {"hidden_dim": 14, "log2_hashmap_size": 16, "num_levels": 4, "max_res": 256, "use_linear": False}],
    proposal_initial_sampler="piecewise", use_single_jitter=True, predict_normals=False, disable_scene_contraction=False,
    use_average_appearance_embedding=True, appearance_embed_dim=29, implementation="torch", camera_optimizer_mode="SO3xR3",
    use_transient_embedding=False, pass_thermal_gradients=True, eval_num_rays_per_chunk=4096, max_temperature=1.0,
    min_temperature=0.0, cold=False,
)
CONFIG_VARIANTS["default"] = dict(_BASE)
CONFIG_VARIANTS["same_proposal_network"] = dict(_BASE, use_same_proposal_network=True,
                                                proposal_net_args_list=[dict(_BASE["proposal_net_args_list"][1])])
CONFIG_VARIANTS["uniform_sampler"] = dict(_BASE, proposal_initial_sampler="uniform", use_single_jitter=False)
CONFIG_VARIANTS["no_contraction"] = dict(_BASE, disable_scene_contraction=True, use_average_appearance_embedding=False,
                                         pass_thermal_gradients=False, camera_optimizer_mode="off")
# more sampling iterations than argument dictionaries: the reference re-uses the last one [REF :136-149]
CONFIG_VARIANTS["three_iterations"] = dict(_BASE, num_proposal_iterations=3, num_proposal_samples_per_ray=(255, 95, 63),
                                           proposal_update_every=7, proposal_warmup=3000)
SCHEDULE_STEPS = (0, 1, 2500, 5000, 10000)
NUM_TRAIN_DATA = 11


class Log:
    """constructor calls in order: [class name, {kwarg: encoded value}]"""

    def __init__(self) -> None:
        self.calls: List[List[Any]] = []
        self.instances: List[Any] = []

    def encode(self, v: Any) -> Any:
        if isinstance(v, Recorded):
            return {"__built__": v._rec_index}
        if isinstance(v, torch.Tensor):
            return {"__tensor__": [float(x) for x in v.reshape(-1).tolist()]}
        if callable(v) and not isinstance(v, type):
            return {"__callable__": getattr(v, "__name__", type(v).__name__)}
        if isinstance(v, float):
            return "inf" if v == float("inf") else v
        if isinstance(v, (list, tuple)):
            return [self.encode(x) for x in v]
        if isinstance(v, dict):
            return {k: self.encode(x) for k, x in v.items()}
        if v is None or isinstance(v, (bool, int, str)):
            return v
        return {"__object__": type(v).__name__}


class Recorded(nn.Module):
    """base of every stand-in: notes its constructor arguments in the shared log"""

    _rec_name = "?"
    _rec_log: Log = None  # type: ignore[assignment]

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        super().__init__()
        log = type(self)._rec_log
        self._rec_index = len(log.calls)
        log.calls.append([type(self)._rec_name, [log.encode(a) for a in args], {k: log.encode(v) for k, v in sorted(kwargs.items())}])
        log.instances.append(self)
        self._rec_kwargs = kwargs

    def density_fn(self, positions):  # HashMLPDensityField.density_fn is handed to the sampler at call time
        raise NotImplementedError

    def setup(self, **kwargs: Any) -> "Recorded":  # CameraOptimizerConfig.setup(num_cameras=..., device=...)
        cls = recorder("CameraOptimizer", type(self)._rec_log)
        return cls(mode=self._rec_kwargs.get("mode"), **kwargs)


def recorder(name: str, log: Log) -> type:
    return type(name, (Recorded,), {"_rec_name": name, "_rec_log": log})


def summarize(model: nn.Module, log: Log) -> Dict[str, Any]:
    """what is compared between the reference's and the product's populate_modules"""
    attrs = {}
    for k, v in list(model._modules.items()):
        if isinstance(v, Recorded):
            attrs[k] = {"__built__": v._rec_index}
        elif isinstance(v, nn.ModuleList):
            attrs[k] = [{"__built__": m._rec_index} for m in v if isinstance(m, Recorded)]
    sampler = next(m for m in log.instances if type(m)._rec_name == "ProposalNetworkSampler")
    sched = sampler._rec_kwargs["update_sched"]
    density_fns = [fn.__self__._rec_index for fn in model.density_fns]
    return {"calls": log.calls, "attributes": attrs, "density_fns": density_fns,
            "update_schedule": {str(s): float(sched(s)) for s in SCHEDULE_STEPS}}
