"""tn_ssim_fwd and ThermalNerfModel.get_image_metrics_and_images [REF thermal_nerf_model.py:328-393] against the CPU oracle
(oracle/metrics.py: torchmetrics' SSIM / PSNR restated; the reference's own mae_thermal is pinned by golden G2)."""
import copy
import math

import pytest
import torch

from oracle import hotpath as H
from oracle import metrics as OM
from tests import helpers
from thermo_nerf_amd import metrics as M
from thermo_nerf_amd.rendered_image_modalities import RenderedImageModality as RM

pytestmark = pytest.mark.gpu
DEV = "cuda"


def chw(x):  # [H,W,C] -> [1,C,H,W], the view the reference hands to torchmetrics
    return torch.moveaxis(x, -1, 0)[None, ...]


@pytest.mark.parametrize("shape", [(11, 11, 1), (23, 17, 3), (64, 48, 1), (135, 240, 3), (43, 75, 4), (800, 800, 3)])
def test_ssim_matches_oracle(shape):
    g = torch.Generator().manual_seed(sum(shape))
    gt = torch.rand(shape, generator=g)
    pred = (gt + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
    got = float(M.ssim(pred.to(DEV), gt.to(DEV)))
    want = float(OM.ssim(chw(pred), chw(gt)))
    assert abs(got - want) <= 1e-5, (got, want)


def test_ssim_edge_cases():
    g = torch.Generator().manual_seed(1)
    a = torch.rand(40, 52, 3, generator=g).to(DEV)
    assert float(M.ssim(a, a)) == pytest.approx(1.0, abs=1e-6)
    b = torch.rand(40, 52, 3, generator=g).to(DEV)
    assert float(M.ssim(a, b)) == pytest.approx(float(M.ssim(b, a)), abs=1e-6)
    # images with a small dynamic range (a thermal frame of a nearly uniform scene): data_range follows the data, as
    # torchmetrics' default does.  The window variances are E[x^2] - E[x]^2 in fp32 — the published algorithm — so the index
    # loses digits as the contrast shrinks (at 0.4 +- 0.005 two fp32 evaluations differ by 5e-3): a looser bound here
    lo, lo2 = 0.4 + 0.1 * a[..., :1], 0.4 + 0.1 * b[..., :1]
    assert abs(float(M.ssim(lo, lo2)) - float(OM.ssim(chw(lo.cpu()), chw(lo2.cpu())))) <= 2e-4
    # a non-contiguous view is accepted (made contiguous), a CPU tensor and a too-small image are refused
    assert float(M.ssim(a[:, ::2], a[:, ::2])) == pytest.approx(1.0, abs=1e-6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.ssim(a.cpu(), a.cpu())
    with pytest.raises(ValueError, match="11 x 11"):
        M.ssim(a[:10], a[:10])
    with pytest.raises(ValueError):
        M.ssim(a, a[:20])


def test_get_image_metrics_and_images():
    """Keys, shapes and values of the reference's per-frame dictionaries on a rendered 36 x 64 frame."""
    model, sd, ocfg = helpers.build("scene", 48)
    gm = copy.deepcopy(model).to(DEV).eval()
    gm.max_temperature, gm.min_temperature = 33.085, 13.896
    o, d = helpers.rays(36, 64, view=1)
    want = H.get_outputs(sd, o, d, None, ocfg)
    from thermo_nerf_amd.rays import RayBundle

    with torch.no_grad():
        out = gm.get_outputs_for_camera_ray_bundle(RayBundle(origins=o.view(36, 64, 3).to(DEV), directions=d.view(36, 64, 3).to(DEV)))
    g = torch.Generator().manual_seed(4)
    gt_rgb = (want["rgb"].view(36, 64, 3) + 0.05 * torch.randn(36, 64, 3, generator=g)).clamp(0, 1)
    gt_th = (want["thermal"].view(36, 64, 1) + 0.05 * torch.randn(36, 64, 1, generator=g)).clamp(0, 1)
    batch = {"image": gt_rgb, RM.THERMAL.value: gt_th}  # host tensors, as a dataloader hands them over
    thr = float(gt_th.median())
    metrics, images = gm.get_image_metrics_and_images(out, batch, threshold=thr)
    assert set(metrics) == {"psnr", "ssim", "lpips", "psnr_thermal", "ssim_thermal", "lpips_thermal", "mae_thermal_foreground",
                            "mae_thermal"}
    rgb, th = out["rgb"].cpu(), out["thermal"].cpu()
    assert metrics["psnr"] == pytest.approx(float(OM.psnr(gt_rgb, rgb)), abs=1e-3)
    assert metrics["psnr_thermal"] == pytest.approx(float(OM.psnr(gt_th, th)), abs=1e-3)
    assert metrics["ssim"] == pytest.approx(float(OM.ssim(chw(gt_rgb), chw(rgb))), abs=1e-5)
    assert metrics["ssim_thermal"] == pytest.approx(float(OM.ssim(chw(gt_th), chw(th))), abs=1e-5)
    assert math.isnan(metrics["lpips"]) and math.isnan(metrics["lpips_thermal"])
    span = 33.085 - 13.896
    assert metrics["mae_thermal"] == pytest.approx(float((gt_th - th).abs().mean()) * span, rel=1e-4)
    fg = gt_th > thr
    assert 0 < int(fg.sum()) < fg.numel()
    assert metrics["mae_thermal_foreground"] == pytest.approx(float((gt_th - th)[fg].abs().mean()) * span, rel=1e-4)
    assert set(images) == {"img", "accumulation", "depth", "thermal", "thermal_combined", "prop_depth_0", "prop_depth_1"}
    assert images["img"].shape == (36, 128, 3) and images["thermal_combined"].shape == (36, 128, 3)
    for k in ("accumulation", "depth", "thermal", "prop_depth_0", "prop_depth_1"):
        assert images[k].shape == (36, 64, 3) and float(images[k].min()) >= -1e-6 and float(images[k].max()) <= 1.0 + 1e-6
    assert torch.equal(images["thermal"][..., 0], out["thermal"][..., 0])  # NS "gray": the value on three channels
    assert torch.equal(images["img"][:, 64:], out["rgb"])
    # an RGBA ground truth is composited over black before it is scored (NS RGBRenderer.blend_background)
    rgba = torch.cat([gt_rgb, torch.full((36, 64, 1), 0.5)], dim=-1)
    m2, _ = gm.get_image_metrics_and_images(out, {"image": rgba, RM.THERMAL.value: gt_th})
    assert m2["psnr"] == pytest.approx(float(OM.psnr(gt_rgb * 0.5, rgb)), abs=1e-3)


def test_thermal_image_metrics_golden_g7(golden_dir):
    """ThermalNerfModel.get_image_metrics_and_images against what the reference's own method returned for the thermal modality
    (tests/golden/model_wiring.npz ``metrics.*``, tools/make_golden.py G7)."""
    import os

    import numpy as np

    g = np.load(os.path.join(golden_dir, "model_wiring.npz"))
    gt, pr = torch.from_numpy(g["metrics.gt_thermal"]).to(DEV), torch.from_numpy(g["metrics.pred_thermal"]).to(DEV)
    cm, _, _ = helpers.build("scene", 48)
    gm = copy.deepcopy(cm).to(DEV).eval()
    gm.config = copy.deepcopy(gm.config)
    gm.max_temperature, gm.min_temperature = (float(x) for x in g["metrics.bounds"])
    Hh, Ww = gt.shape[:2]
    out = {"rgb": torch.rand(Hh, Ww, 3, device=DEV), "accumulation": torch.rand(Hh, Ww, 1, device=DEV),
           "depth": torch.rand(Hh, Ww, 1, device=DEV) + 1, "thermal": pr}
    batch = {"image": torch.rand(Hh, Ww, 3, device=DEV), "thermal": gt}
    for cold in (False, True):
        gm.config.cold = cold
        for thr in (None, 0.6):
            want = dict(zip(g[f"metrics.cold{int(cold)}_thr{thr}.keys"].tolist(), g[f"metrics.cold{int(cold)}_thr{thr}.values"].tolist()))
            m, im = gm.get_image_metrics_and_images(out, batch, threshold=thr)
            assert list(m.keys()) == list(want.keys())
            for k in ("psnr_thermal", "ssim_thermal", "mae_thermal_foreground", "mae_thermal"):
                assert abs(m[k] - want[k]) <= 2e-5 * abs(want[k]) + 1e-6, (k, m[k], want[k])
    assert [k for k in im if not k.startswith("prop_depth")] == g["metrics.image_keys"].tolist()
    np.testing.assert_array_equal(im["thermal"].cpu().numpy(), g["metrics.thermal_image"])
    np.testing.assert_array_equal(im["thermal_combined"].cpu().numpy(), g["metrics.thermal_combined_image"])
