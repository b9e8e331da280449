"""CPU: pin the oracle against the golden fixtures (G1/G2 come from the REAL reference modules; G3 is the
oracle's own regression set) — SURVEY §8c."""
import os

import numpy as np
import torch

from oracle import hotpath as H
from thermo_nerf_amd.thermal_nerf.thermal_metrics import mae_thermal as product_mae


def test_g1_thermal_renderer_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "thermal_renderer.npz"))
    th, w = torch.from_numpy(g["thermal"]), torch.from_numpy(g["weights"])
    out_train = H.render_thermal(th, w, training=True)
    out_eval = H.render_thermal(th, w, training=False)
    np.testing.assert_array_equal(np.isnan(out_train.numpy()), np.isnan(g["out_train"]))
    fin = np.isfinite(g["out_train"])
    np.testing.assert_allclose(out_train.numpy()[fin], g["out_train"][fin], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(np.isinf(out_train.numpy()), np.isinf(g["out_train"]))
    np.testing.assert_allclose(out_eval.numpy(), g["out_eval"], rtol=0, atol=1e-6)
    assert np.all(g["out_eval"] >= 0) and np.all(g["out_eval"] <= 1)


def test_g2_mae_thermal_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "mae_thermal.npz"))
    gt, pred = torch.from_numpy(g["gt"]), torch.from_numpy(g["pred"])
    tmax, tmin = float(g["tmax"]), float(g["tmin"])
    for cold in (False, True):
        for thr in (None, 0.5):
            want = float(g[f"cold{int(cold)}_thr{thr}"])
            assert abs(float(H.mae_thermal(gt, pred, cold, tmax, tmin, thr)) - want) < 1e-6
            assert abs(float(product_mae(gt, pred, cold, tmax, tmin, thr)) - want) < 1e-6


def test_g3_oracle_regression(golden_dir):
    g = np.load(os.path.join(golden_dir, "oracle_regression.npz"))
    for L, lo, hi in ((16, 16, 2048), (5, 16, 128), (5, 16, 256)):
        np.testing.assert_array_equal(H.hash_scalings(L, lo, hi).numpy(), g[f"scalings_{L}_{lo}_{hi}"])
    # the values SURVEY §7 probed in float32
    assert g["scalings_16_16_2048"].tolist() == [16, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047]
    coords = torch.from_numpy(g["hash_coords"]).view(1000, 1, 3).expand(-1, 5, -1).contiguous()
    np.testing.assert_array_equal(H.hash_fn(coords, 2**17, torch.arange(5) * 2**17).numpy(), g["hash_idx_T17_L5"])
    np.testing.assert_allclose(H.contract_inf(torch.from_numpy(g["contract_in"])).numpy(), g["contract_out"], atol=0)


def test_hash_low_bits_equal_uint32_wrap(golden_dir):
    """The kernels hash in wrapping uint32; the reference in int64.  Low log2T bits must agree."""
    g = np.load(os.path.join(golden_dir, "oracle_regression.npz"))
    c = g["hash_coords"].astype(np.uint32)
    h32 = (c[:, 0] ^ (c[:, 1] * np.uint32(2654435761)) ^ (c[:, 2] * np.uint32(805459861))) & np.uint32(2**17 - 1)
    np.testing.assert_array_equal(h32.astype(np.int64), g["hash_idx_T17_L5"][:, 0])


def _close_with_nonfinite(got, want, atol=1e-6):
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(np.isposinf(got), np.isposinf(want))
    np.testing.assert_array_equal(np.isneginf(got), np.isneginf(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=atol)


def test_g4_rgb_renderer_matches_reference_rgbt_renderer(golden_dir):
    """G4: the reference's own fork of nerfstudio's RGBRenderer (rgb_concat/rgbt_renderer.py:62-81,159-174), background
    "last_sample": pins ``H.render_rgb`` in train and eval mode, on 3 and 4 channels."""
    g = np.load(os.path.join(golden_dir, "rgbt_renderer.npz"))
    rgbt, w = torch.from_numpy(g["rgbt"]), torch.from_numpy(g["weights"])
    for mode in ("train", "eval"):
        _close_with_nonfinite(H.render_rgb(rgbt, w, training=(mode == "train")).numpy(), g[f"out4_{mode}"])
        _close_with_nonfinite(H.render_rgb(rgbt[..., :3], w, training=(mode == "train")).numpy(), g[f"out3_{mode}"])
    assert np.all(g["out3_eval"] >= 0) and np.all(g["out3_eval"] <= 1)
    # the thermal compositor is the same arithmetic on one channel [REF thermal_renderer.py:27-80]
    _close_with_nonfinite(H.render_thermal(rgbt[..., 3:], w, training=False).numpy(), g["out4_eval"][:, 3:])


def test_g5_thermal_field_head_matches_reference(golden_dir):
    """G5: BaseThermalFieldHead (thermal_field_head.py:50-71) = Linear(64, 1), no activation; parameter names net.*."""
    g = np.load(os.path.join(golden_dir, "thermal_field_head.npz"))
    y = torch.nn.functional.linear(torch.from_numpy(g["x"]), torch.from_numpy(g["weight"]), torch.from_numpy(g["bias"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=1e-6)
    assert list(g["state_keys"]) == ["net.bias", "net.weight"]
    assert str(g["enum_value"]) == "thermal"
    from thermo_nerf_amd.thermal_nerf.thermal_field_head import FieldHeadNamesT, ThermalFieldHead

    assert FieldHeadNamesT.THERMAL.value == str(g["enum_value"])
    assert sorted(ThermalFieldHead(in_dim=64).state_dict().keys()) == list(g["state_keys"])


def _wiring_fixture(golden_dir):
    import json

    g = np.load(os.path.join(golden_dir, "thermal_field_wiring.npz"))
    cfgd = json.loads(str(g["config"]))
    sd = {"field." + k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    ocfg = H.OracleConfig(num_levels=cfgd["num_levels"], log2_hashmap_size=cfgd["log2_hashmap_size"],
                          base_res=cfgd["base_res"], max_res=cfgd["max_res"])  # scene contraction on (reference default)
    return g, sd, ocfg


def test_g6_field_wiring_matches_reference_thermal_field(golden_dir):
    """G6: outputs of the REAL ThermalNerfactoTField.forward (thermal_field.py:108-201) run over oracle-built nerfstudio base
    classes.  The oracle's own ``field_density`` + ``field_outputs`` must reproduce them: this pins the reference file's
    wiring (argument order into NerfactoField, concat order, appearance branches, head without activation), not
    nerfstudio's arithmetic."""
    g, sd, ocfg = _wiring_fixture(golden_dir)
    pos, dirs, cam = (torch.from_numpy(g[k]) for k in ("positions", "directions", "camera_indices"))
    for avg in (1, 0):
        ocfg.use_average_appearance_embedding = bool(avg)
        density, geo = H.field_density(sd, pos, ocfg)
        for mode in ("eval", "train"):
            rgb, th = H.field_outputs(sd, dirs, geo, cam, ocfg, training=(mode == "train"))
            tag = f"avg{avg}_{mode}"
            np.testing.assert_allclose(density.numpy(), g[f"density_{tag}"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(rgb.numpy(), g[f"rgb_{tag}"], rtol=0, atol=1e-6)
            np.testing.assert_allclose(th.numpy(), g[f"thermal_{tag}"], rtol=0, atol=1e-6)
    # eval with the average embedding differs from train (per-camera embedding), and avg0 eval uses zeros
    assert np.abs(g["rgb_avg1_eval"] - g["rgb_avg1_train"]).max() > 1e-4
    assert np.abs(g["rgb_avg1_eval"] - g["rgb_avg0_eval"]).max() > 1e-4
    np.testing.assert_array_equal(g["thermal_avg1_eval"], g["thermal_avg1_train"])  # thermal never sees the embedding
    assert str(g["missing_cam_message"]) == "Camera indices are not provided."


def test_g6_state_dict_names_match_reference_module(golden_dir):
    """The reference module's own parameter names (mlp_thermal.*, field_head_thermal.net.*) and the nerfstudio names its
    base class contributes are the product's state-dict names: a nerfstudio checkpoint loads unchanged."""
    from thermo_nerf_amd.thermal_nerf.thermal_field import ThermalNerfactoTField

    g = np.load(os.path.join(golden_dir, "thermal_field_wiring.npz"))
    ref_keys = {k for k in g["state_keys"] if not k.endswith("aabb")}
    f = ThermalNerfactoTField(torch.tensor([[-1.0] * 3, [1.0] * 3]), num_images=5, log2_hashmap_size=10)
    mine = set(f.state_dict().keys())
    assert ref_keys <= mine, sorted(ref_keys - mine)
    for k in ref_keys:
        assert tuple(f.state_dict()[k].shape) == tuple(g["sd." + k].shape), k


# ---- G7: the reference's own get_outputs / get_loss_dict on oracle-built components ----------------------------------------------
def g7_problem(golden_dir):
    """(fixture, cpu model, state dict incl. the fixture's pose adjustment, oracle config, rays, cameras, jitter, batch)"""
    import json

    from tests import helpers

    g = np.load(os.path.join(golden_dir, "model_wiring.npz"))
    cfg = json.loads(str(g["config"]))
    cm, sd, ocfg = helpers.build(cfg["kind"], cfg["S"], small=cfg["small"], num_images=cfg["num_images"],
                                 camera_optimizer_mode=cfg["camera_optimizer_mode"], log2_hashmap_size=cfg["log2_hashmap_size"],
                                 num_proposal_samples_per_ray=tuple(cfg["num_proposal_samples_per_ray"]))
    sd = {k: v.clone() for k, v in sd.items()}
    sd["camera_optimizer.pose_adjustment"] = torch.from_numpy(g["pose_adjustment"])
    # the weights are regenerated, not stored: they must be the tensors the fixture was made from
    assert sorted(sd.keys()) == g["sd_keys"].tolist()
    sums = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in sorted(sd.keys())])
    np.testing.assert_allclose(sums, g["sd_sums"], rtol=1e-12, atol=0)
    o, d = torch.from_numpy(g["origins"]), torch.from_numpy(g["directions"])
    cam = torch.from_numpy(g["camera_indices"])
    jit = [j for j in torch.from_numpy(g["jitter"])]
    batch = {"image": torch.from_numpy(g["image"]), "thermal": torch.from_numpy(g["thermal"])}
    return g, cm, sd, ocfg, o, d, cam, jit, batch


G7_CASES = {"eval": dict(training=False), "train": dict(training=True), "train_scaled": dict(training=True, gradient_scaling=True),
            "train_no_thermal": dict(training=True, pass_thermal=False)}
G7_GRADS = ["field.mlp_base.mlp.layers.1.weight", "field.mlp_thermal.layers.0.weight", "field.mlp_head.layers.0.weight",
            "proposal_networks.0.mlp_base.mlp.layers.0.weight", "camera_optimizer.pose_adjustment"]


def test_g7_get_outputs_and_get_loss_dict_follow_the_reference_methods(golden_dir):
    """oracle.hotpath.get_outputs + oracle.training.get_loss_dict against what the reference's OWN ThermalNerfModel.get_outputs /
    get_loss_dict [REF thermal_nerf_model.py:210-326] returned when run over oracle-built components (tools/make_golden.py G7):
    output keys and their order, train-only lists, prop_depth_i, camera optimizer in training only, loss keys / multipliers /
    gates, gradient scaling.  Pins the wiring of the two methods, not nerfstudio's arithmetic."""
    import dataclasses

    from oracle import training as T

    g, cm, sd, ocfg0, o, d, cam, jit, batch = g7_problem(golden_dir)
    assert g["train.call_log"].tolist() == ["camera_optimizer", "proposal_sampler", "field", "renderer_rgb"]
    assert g["eval.call_log"].tolist() == ["proposal_sampler", "field", "renderer_rgb"]  # REF :218: training only
    for tag, kw in G7_CASES.items():
        training = kw["training"]
        ocfg = dataclasses.replace(ocfg0, use_gradient_scaling=bool(kw.get("gradient_scaling", False)))
        leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith((".aabb", ".scalings")) else v)
                  for k, v in sd.items()}
        oo, dd = (T.apply_pose_adjustment(leaves["camera_optimizer.pose_adjustment"], cam, o, d) if training else (o, d))
        out = H.get_outputs(leaves, oo, dd, cam if training else None, ocfg, training=training, jitter=jit if training else None)
        assert list(out.keys()) == g[f"{tag}.output_keys"].tolist(), tag
        for k, v in out.items():
            if isinstance(v, torch.Tensor):
                np.testing.assert_allclose(v.detach().numpy(), g[f"{tag}.out.{k}"], rtol=0, atol=1e-6, err_msg=f"{tag} {k}")
        metrics = T.get_metrics_dict(out, batch, training)
        loss = T.get_loss_dict(out, batch, metrics, training, pass_thermal_gradients=kw.get("pass_thermal", True))
        assert list(loss.keys()) == g[f"{tag}.loss_keys"].tolist(), tag
        for k, v in loss.items():
            assert abs(v.item() - float(g[f"{tag}.loss.{k}"])) <= 1e-6 * abs(v.item()) + 1e-9, (tag, k)
        assert g[f"{tag}.calls"].tolist() == [1 if training else 0, 1 if kw.get("gradient_scaling") else 0]
        if training:
            for i in range(3):
                np.testing.assert_allclose(out["weights_list"][i].detach().numpy(), g[f"{tag}.out.weights_list.{i}"], atol=1e-6)
                np.testing.assert_allclose(T.ray_samples_to_sdist(out["ray_samples_list"][i]).numpy(), g[f"{tag}.out.spacing_bins.{i}"],
                                           atol=1e-6)
            sum(loss.values()).backward()
            for name in G7_GRADS:
                want = g[f"{tag}.grad.{name}"]
                got = leaves[name].grad
                got = np.zeros_like(want) if got is None else got.numpy()
                np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-9 + 1e-5 * np.abs(want).max(), err_msg=f"{tag} {name}")
    # the gates do what the reference's method did: no thermal loss -> the thermal MLP receives nothing; scaling changes gradients
    assert np.abs(g["train_no_thermal.grad.field.mlp_thermal.layers.0.weight"]).max() == 0.0
    assert np.abs(g["train.grad.field.mlp_thermal.layers.0.weight"]).max() > 0.0
    assert not np.allclose(g["train_scaled.grad.field.mlp_base.mlp.layers.1.weight"], g["train.grad.field.mlp_base.mlp.layers.1.weight"])


def test_g7_thermal_image_metrics_follow_the_reference_method(golden_dir):
    """oracle.metrics.thermal_image_metrics against the reference's own get_image_metrics_and_images [REF :328-393] (run with
    the oracle's PSNR / SSIM as the torchmetrics stand-ins and the reference's own mae_thermal): metric keys and their order,
    [1,C,H,W] layout, which MAE takes the threshold, the "gray" thermal images."""
    from oracle import metrics as OM

    g = np.load(os.path.join(golden_dir, "model_wiring.npz"))
    gt, pr = torch.from_numpy(g["metrics.gt_thermal"]), torch.from_numpy(g["metrics.pred_thermal"])
    tmax, tmin = (float(x) for x in g["metrics.bounds"])
    for cold in (False, True):
        for thr in (None, 0.6):
            keys = g[f"metrics.cold{int(cold)}_thr{thr}.keys"].tolist()
            vals = dict(zip(keys, g[f"metrics.cold{int(cold)}_thr{thr}.values"].tolist()))
            assert keys == ["psnr", "ssim", "lpips", "psnr_thermal", "ssim_thermal", "lpips_thermal", "mae_thermal_foreground",
                            "mae_thermal"]
            got = OM.thermal_image_metrics(gt, pr, cold, tmax, tmin, thr)
            for k, v in got.items():
                assert abs(v - vals[k]) <= 1e-6 * abs(vals[k]) + 1e-7, (k, v, vals[k])
    assert g["metrics.image_keys"].tolist() == ["img", "accumulation", "depth", "thermal", "thermal_combined"]
    np.testing.assert_array_equal(g["metrics.thermal_image"], np.repeat(g["metrics.pred_thermal"], 3, axis=-1))
    np.testing.assert_array_equal(g["metrics.thermal_combined_image"],
                                  np.concatenate([np.repeat(g["metrics.gt_thermal"], 3, -1), np.repeat(g["metrics.pred_thermal"], 3, -1)], axis=1))
    assert g["metrics.lpips_input_shape"].tolist() == [1, 3, 24, 20]  # the thermal channel repeated to three for LPIPS [REF :381-383]
