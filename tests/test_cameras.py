"""Camera path loading (CPU, against the reference's own trajectory fixture) and on-device ray generation (GPU,
against the oracle restatement of nerfstudio's Cameras.generate_rays)."""
import json
import math
import os

import pytest
import torch

from oracle import cameras as OC
from thermo_nerf_amd import cameras as C


def test_camera_path_fixture_loads_96_cameras(golden_dir):
    """The reference's own assertion on this file: 96 cameras [REF tests/test_renderer.py:65-69]."""
    cams = C.load_cameras(os.path.join(golden_dir, "camera_path_facade_2.json"))
    assert cams.size == 96 and len(cams) == 96
    assert (cams.height, cams.width) == (1080, 1920)
    assert cams.camera_to_worlds.shape == (96, 3, 4)
    want_f = 540.0 / math.tan(math.radians(50.0) / 2.0)
    assert abs(float(cams.fx[0]) - want_f) < 1e-3 and float(cams.fx[0]) == float(cams.fy[0])
    assert (cams.cx, cams.cy) == (960.0, 540.0)
    half = C.load_cameras(os.path.join(golden_dir, "camera_path_facade_2.json"), 0.5)
    assert (half.height, half.width) == (540, 960) and abs(float(half.fx[0]) - want_f / 2) < 1e-3


def test_frame_metrics_uses_reference_mae(golden_dir):
    import numpy as np
    g = np.load(os.path.join(golden_dir, "mae_thermal.npz"))
    gt, pred = torch.from_numpy(g["gt"])[0, 0, :, :, None], torch.from_numpy(g["pred"])[0, 0, :, :, None]
    m = C.frame_metrics({"rgb": pred.repeat(1, 1, 3), "thermal": pred}, gt.repeat(1, 1, 3), gt, float(g["tmax"]),
                        float(g["tmin"]), cold=False, threshold=0.5)
    assert abs(m["mae_thermal"] - float(g["cold0_thrNone"])) < 1e-5
    assert abs(m["mae_thermal_foreground"] - float(g["cold0_thr0.5"])) < 1e-5
    assert abs(m["psnr"] - m["psnr_thermal"]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [0, 37, 95])
def test_generate_rays_matches_oracle(golden_dir, idx):
    cams = C.load_cameras(os.path.join(golden_dir, "camera_path_facade_2.json"), 0.125)  # 135 x 240
    rb = cams.generate_rays(idx, device="cuda:0")
    o, d, a = OC.generate_rays(cams.camera_to_worlds[idx], float(cams.fx[idx]), float(cams.fy[idx]), cams.cx, cams.cy,
                               cams.height, cams.width)
    assert rb.origins.shape == (cams.height, cams.width, 3)
    assert torch.equal(rb.origins.cpu(), o.contiguous())
    assert (rb.directions.cpu() - d).abs().max().item() < 2e-7
    assert ((rb.pixel_area.cpu() - a).abs() / a).max().item() < 2e-3  # difference of nearly equal unit vectors
    assert (rb.directions.norm(dim=-1) - 1).abs().max().item() < 1e-6
    assert int(rb.camera_indices[0, 0, 0]) == idx
    # a row block equals the same rows of the full frame (what each rank generates when a frame is sharded)
    part = cams.generate_rays(idx, device="cuda:0", rows=(40, 77))
    assert torch.equal(part.directions, rb.directions[40:77])


@pytest.mark.gpu
def test_render_camera_path_frame_end_to_end(golden_dir):
    """cameras -> rays on device -> model.get_outputs_for_camera_ray_bundle == oracle on the same camera."""
    import copy
    from oracle import hotpath as H
    from tests import helpers

    cams = C.load_cameras(os.path.join(golden_dir, "camera_path_facade_2.json"), 1.0 / 30)  # 36 x 64
    model, sd, ocfg = helpers.build("scene", 48)
    gm = copy.deepcopy(model).to("cuda:0").eval()
    gm.config.eval_num_rays_per_chunk = 1000
    rb = cams.generate_rays(5, device="cuda:0")
    got = gm.get_outputs_for_camera_ray_bundle(rb)
    o, d, _ = OC.generate_rays(cams.camera_to_worlds[5], float(cams.fx[5]), float(cams.fy[5]), cams.cx, cams.cy,
                               cams.height, cams.width)
    want = H.get_outputs_for_camera_ray_bundle(sd, o.contiguous(), d.contiguous(), ocfg, chunk=1000)
    assert got["rgb"].shape == (cams.height, cams.width, 3)
    assert (got["rgb"].cpu() - want["rgb"]).abs().mean().item() < 1e-4
    assert (got["thermal"].cpu() - want["thermal"]).abs().mean().item() < 1e-4
    m = C.frame_metrics(got, want["rgb"], want["thermal"], 33.085, 13.896)
    assert m["psnr"] > 60 and m["mae_thermal"] < 2e-3  # degrees C over a 19.2 C span


def test_undistortion_oracle_inverts_the_opencv_model():
    """CPU: NS radial_and_tangential_undistort restated in oracle/cameras.py is the inverse of the forward model."""
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(500, 2, generator=g) - 0.5) * 1.2
    for k in ([0.12, -0.05, 0.0, 0.0, 0.003, -0.002], [-0.2, 0.04, 0.01, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.01, 0.01]):
        kk = torch.tensor(k)
        und = OC.radial_and_tangential_undistort(OC.distort(pts, kk), kk)
        assert (und - pts).abs().max().item() < 1e-5, k
    zero = torch.zeros(6)
    assert torch.equal(OC.radial_and_tangential_undistort(pts, zero), pts)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [[0.12, -0.05, 0.0, 0.0, 0.003, -0.002], [-0.25, 0.08, 0.0, 0.0, 0.0, 0.0],
                               [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]])
def test_generate_rays_with_lens_distortion_matches_oracle(k):
    """OPENCV k1/k2/p1/p2 as nerfstudio-processed ThermoScenes carry them: device rays vs the oracle's Newton undistortion."""
    from thermo_nerf_amd import synthetic

    cams = synthetic.orbit_cameras(60, 80, [0, 3], num_views=8)
    cams.distortion_params = torch.tensor([k, k])
    rb = cams.generate_rays(1, device="cuda:0")
    o, d, a = OC.generate_rays(cams.camera_to_worlds[1], float(cams.fx[1]), float(cams.fy[1]), cams.cx, cams.cy, 60, 80,
                               distortion_params=torch.tensor(k))
    assert torch.equal(rb.origins.cpu(), o)
    assert (rb.directions.cpu() - d).abs().max().item() <= 2e-6
    assert ((rb.pixel_area.cpu() - a).abs() / a).max().item() <= 2e-3
    plain = OC.generate_rays(cams.camera_to_worlds[1], float(cams.fx[1]), float(cams.fy[1]), cams.cx, cams.cy, 60, 80)[1]
    if any(k):
        assert (d - plain).abs().max().item() > 1e-3  # the distortion actually bends the corner rays
    else:
        assert torch.equal(d, plain)


def test_random_pixel_batch_draws_the_same_rays_as_the_orbit_images():
    """synthetic.random_pixel_rays (the PixelSampler-style training batch of bench.py / tools/train_bench.py) against the
    per-image ray generator at the drawn pixels: same origins, same directions, camera index = the view."""
    from thermo_nerf_amd import synthetic

    n, H, W = 512, 80, 60
    o, d, cam = synthetic.random_pixel_rays(n, H, W, num_views=8, seed=3)
    assert o.shape == (n, 3) and d.shape == (n, 3) and cam.shape == (n, 1) and cam.dtype == torch.int64
    g = torch.Generator().manual_seed(3)
    view = torch.randint(0, 8, (n,), generator=g)
    ys = torch.randint(0, H, (n,), generator=g)
    xs = torch.randint(0, W, (n,), generator=g)
    assert torch.equal(cam[:, 0], view)
    assert len(view.unique()) == 8  # every image contributes
    for v in range(8):
        oi, di, _ = synthetic.orbit_camera_rays(H, W, view=v)
        m = view == v
        assert torch.allclose(o[m], oi[ys[m], xs[m]], atol=0, rtol=0)
        assert torch.allclose(d[m], di[ys[m], xs[m]], atol=1e-6, rtol=0)
