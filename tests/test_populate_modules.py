"""G8: the product's ``ThermalNerfModel.populate_modules`` / method config against what the REFERENCE's own code does when
executed (tests/golden/populate_modules.json, written by tools/make_golden_g8.py from /root/reference/thermo_nerf/thermal_nerf/
thermal_nerf_model.py:67-208, nerfacto_config/thermal_nerfacto.py:31-45 and thermal_nerf/config_thermal_nerf.py:17-48 with
nerfstudio's constructors replaced by recorders).  Both sides run through the same recording stand-ins (tests/g8_harness.py):
which class is built, in which order, from which config value; which attribute holds it; ``density_fns``; the proposal
update schedule."""
import json
import os

import pytest

from tests import g8_harness as G

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "populate_modules.json")

# built by the reference, not by the product, and why
NOT_BUILT = {
    "NormalsRenderer": "predict_normals=True ends in KeyError(PRED_NORMALS) in the reference itself (G9, tests/golden/predict_normals.json): never used",
    "NormalsShader": "same",
    "MSELoss": "both MSE terms come from tn_image_losses (one launch with the PSNR)",
    "PeakSignalNoiseRatio": "thermo_nerf_amd.cameras.psnr / tn_image_losses",
    "LearnedPerceptualImagePatchSimilarity": "no pretrained network offline (DESIGN 9)",
}


def canonical(calls):
    """calls with the NOT_BUILT classes dropped and {"__built__": index} replaced by [class name, ordinal among that class]"""
    ordinal, seen = {}, {}
    for i, (name, _, _) in enumerate(calls):
        ordinal[i] = [name, seen.get(name, 0)]
        seen[name] = seen.get(name, 0) + 1

    def fix(v):
        if isinstance(v, dict) and "__built__" in v:
            return {"__built__": ordinal[v["__built__"]]}
        if isinstance(v, dict):
            return {k: fix(x) for k, x in v.items()}
        if isinstance(v, list):
            return [fix(x) for x in v]
        return v

    return [[n, fix(a), fix(k)] for n, a, k in calls if n not in NOT_BUILT], fix


def product_summary(variant):
    import torch

    from thermo_nerf_amd import SceneBox
    from thermo_nerf_amd.nerfacto_config import thermal_nerfacto as base
    from thermo_nerf_amd.thermal_nerf import thermal_nerf_model as M

    log = G.Log()
    mp = pytest.MonkeyPatch()
    try:
        for name in ("ThermalNerfactoTField", "HashMLPDensityField", "ProposalNetworkSampler", "UniformSampler", "NearFarCollider",
                     "RGBRenderer", "AccumulationRenderer", "DepthRenderer", "ThermalRenderer", "SceneContraction"):
            mp.setattr(M, name, G.recorder(name, log))
        mp.setattr(base, "CameraOptimizerConfig", G.recorder("CameraOptimizerConfig", log))
        cfg = M.ThermalNerfModelConfig(**variant)
        with pytest.raises(ValueError, match="Thermal images not found in metadata."):
            M.ThermalNerfModel(cfg, metadata={}, scene_box=SceneBox.unit(), num_train_data=G.NUM_TRAIN_DATA)
        log.calls.clear()
        log.instances.clear()
        model = M.ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]])),
                                   num_train_data=G.NUM_TRAIN_DATA)
        out = G.summarize(model, log)
        out["model_attributes"] = {"max_temperature": model.max_temperature, "min_temperature": model.min_temperature, "step": model.step}
        return out
    finally:
        mp.undo()


@pytest.mark.parametrize("name", list(G.CONFIG_VARIANTS))
def test_populate_modules_builds_what_the_reference_builds(name):
    want = json.load(open(GOLDEN))["populate_modules"][name]
    got = json.loads(json.dumps(product_summary(G.CONFIG_VARIANTS[name])))  # tuples -> lists, like the fixture
    want_calls, want_fix = canonical(want["calls"])
    got_calls, got_fix = canonical(got["calls"])
    assert [c[0] for c in got_calls] == [c[0] for c in want_calls]  # the same classes in the same order
    for (n, a_got, k_got), (_, a_want, k_want) in zip(got_calls, want_calls):
        k_got = dict(k_got)
        if n in ("ThermalNerfactoTField", "HashMLPDensityField"):
            # the one deliberate difference: whatever `implementation` the config names, the fields are built for the HIP
            # kernels ("the build adds one value to implementation"); the reference forwards the config's value
            assert k_want["implementation"] == G.CONFIG_VARIANTS[name]["implementation"] and k_got.pop("implementation") == "hip"
            k_want = {k: v for k, v in k_want.items() if k != "implementation"}
        if n == "ThermalNerfactoTField":
            assert k_got.pop("sh_input") == "shifted"  # an addition of this implementation (SURVEY A.6 switch)
        assert a_got == a_want and k_got == k_want, (n, k_got, k_want)
    # which attribute holds which module (ModuleList order included), and which networks serve the sampler's levels
    want_attrs = {k: want_fix(v) for k, v in want["attributes"].items()
                  if not (isinstance(v, dict) and want["calls"][v["__built__"]][0] in NOT_BUILT)}
    assert {k: got_fix(v) for k, v in got["attributes"].items()} == want_attrs
    assert [got_fix({"__built__": i}) for i in got["density_fns"]] == [want_fix({"__built__": i}) for i in want["density_fns"]]
    assert got["update_schedule"] == want["update_schedule"]
    assert got["model_attributes"] == want["model_attributes"]


def test_method_config_and_default_optimizers_match_the_reference_file():
    from thermo_nerf_amd.thermal_nerf.config_thermal_nerf import thermal_nerf_config as mine
    from thermo_nerf_amd.trainer import TrainerConfig, default_optimizers

    ref = json.load(open(GOLDEN))["method_config"]
    assert mine.method_name == ref["method_name"] == "thermal-nerf"
    assert mine.steps_per_eval_batch == ref["steps_per_eval_batch"]
    for tc in (mine.trainer, TrainerConfig()):  # the method config and the trainer's own defaults
        assert tc.max_num_iterations == ref["max_num_iterations"]
        assert tc.steps_per_save == ref["steps_per_save"]
        assert tc.train_num_rays_per_batch == ref["pipeline"]["datamanager"]["train_num_rays_per_batch"]
    assert mine.trainer.mixed_precision is ref["mixed_precision"] is True
    assert mine.eval_num_rays_per_batch == ref["pipeline"]["datamanager"]["eval_num_rays_per_batch"]
    assert mine.model.eval_num_rays_per_chunk == ref["pipeline"]["model"]["eval_num_rays_per_chunk"] == 1 << 16
    assert set(ref["optimizers"]) == {"proposal_networks", "fields"}
    for table in (mine.trainer.optimizers, default_optimizers()):
        for group, entry in ref["optimizers"].items():
            oc = table[group]
            assert (oc.lr, oc.eps) == (entry["optimizer"]["lr"], entry["optimizer"]["eps"])
            assert (oc.lr_final, oc.max_steps) == (entry["scheduler"]["lr_final"], entry["scheduler"]["max_steps"])
            assert oc.weight_decay == 0.0 and oc.warmup_steps == 0  # nerfstudio's defaults, which the reference leaves alone
