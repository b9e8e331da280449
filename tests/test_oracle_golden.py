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
