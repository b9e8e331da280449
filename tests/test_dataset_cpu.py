"""Dataset boundary: Thermal dataparser + ThermalDataset on a synthetic transforms.json tree and on the
reference's own thermal test image (tests/golden/thermal/, copied data files of REF tests/data/thermal/)."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from thermo_nerf_amd.data import Thermal, ThermalDataParserConfig, ThermalDataset
from thermo_nerf_amd.data.camera_utils import auto_orient_and_center_poses, rotation_matrix

GOLDEN = Path(__file__).parent / "golden" / "thermal"


def make_scene(root: Path, n_train=5, n_eval=2, h=6, w=8, extra=None, per_frame_intrinsics=False, folders=("images", "thermal")):
    (root / folders[0]).mkdir(parents=True)
    (root / folders[1]).mkdir(parents=True)
    rng = np.random.default_rng(0)
    frames = []
    names = [f"frame_train_{i:04d}" for i in range(n_train)] + [f"frame_eval_{i:04d}" for i in range(n_eval)]
    order = rng.permutation(len(names))  # the json lists frames out of order: the parser sorts by file name
    for k in order:
        nm = names[k]
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / folders[0] / f"{nm}.png")
        Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), mode="L").save(root / folders[1] / f"{nm}.png")
        az = 2 * math.pi * k / len(names)
        c2w = np.eye(4)
        c2w[:3, 3] = [3 * math.cos(az) + 10.0, 3 * math.sin(az) - 4.0, 1.0 + 0.1 * k]
        c2w[:3, 1] = [0.0, 0.6, 0.8]  # every camera rotated about x the same way: "up" is tilted off +z
        c2w[:3, 2] = [0.0, -0.8, 0.6]
        fr = {"file_path": f"{folders[0]}/{nm}.png", "thermal_file_path": f"{folders[1]}/{nm}.png", "transform_matrix": c2w.tolist()}
        if per_frame_intrinsics:
            fr.update({"fl_x": 10.0 + k, "fl_y": 11.0 + k, "cx": w / 2, "cy": h / 2, "h": h, "w": w})
        frames.append(fr)
    meta = {"frames": frames}
    if not per_frame_intrinsics:
        meta.update({"fl_x": 10.0, "fl_y": 11.0, "cx": w / 2, "cy": h / 2, "h": h, "w": w})
    meta.update(extra or {})
    (root / "transforms.json").write_text(json.dumps(meta))
    return names


def test_filename_split_sorting_and_thermal_paths(tmp_path):
    make_scene(tmp_path)
    parser = ThermalDataParserConfig(data=tmp_path).setup()
    train, ev = parser.get_dataparser_outputs("train"), parser.get_dataparser_outputs("val")
    assert [p.name for p in train.image_filenames] == [f"frame_train_{i:04d}.png" for i in range(5)]
    assert [p.name for p in ev.image_filenames] == [f"frame_eval_{i:04d}.png" for i in range(2)]
    assert [p.parent.name for p in train.metadata["thermal"]] == ["thermal"] * 5
    assert [p.name for p in train.metadata["thermal"]] == [p.name for p in train.image_filenames]
    assert len(train.cameras) == 5 and len(ev.cameras) == 2
    assert (train.cameras.height, train.cameras.width, train.cameras.cx, train.cameras.cy) == (6, 8, 4.0, 3.0)
    assert torch.equal(train.scene_box.aabb, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))
    with pytest.raises(ValueError):
        parser.get_dataparser_outputs("bogus")
    # a transforms file can be named directly
    assert len(ThermalDataParserConfig(data=tmp_path / "transforms.json").setup().get_dataparser_outputs("test").cameras) == 2


def test_pose_normalisation_is_shared_by_all_splits(tmp_path):
    make_scene(tmp_path)
    parser = ThermalDataParserConfig(data=tmp_path).setup()
    train, ev = parser.get_dataparser_outputs("train"), parser.get_dataparser_outputs("val")
    c2w = torch.cat([train.cameras.camera_to_worlds, ev.cameras.camera_to_worlds])
    t = c2w[:, :3, 3]
    assert torch.allclose(t.mean(dim=0), torch.zeros(3), atol=1e-6)          # centred on the mean camera position
    assert abs(t.abs().max().item() - 1.0) < 1e-6                           # auto-scaled into +-1
    up = c2w[:, :3, 1].mean(dim=0)
    assert torch.allclose(up / up.norm(), torch.tensor([0.0, 0.0, 1.0]), atol=1e-6)  # mean up-vector -> +z
    assert train.dataparser_scale == ev.dataparser_scale and train.dataparser_scale == pytest.approx(1 / 3.0, rel=0.2)
    assert torch.equal(train.dataparser_transform, ev.dataparser_transform)
    # rotations stay rotations
    R = c2w[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand_as(R), atol=1e-5)
    off = ThermalDataParserConfig(data=tmp_path, orientation_method="none", center_method="none", auto_scale_poses=False)
    raw = off.setup().get_dataparser_outputs("train").cameras.camera_to_worlds
    assert raw[:, 0, 3].min().item() > 6.0  # untouched: the +10 offset of the synthetic poses is still there


def test_rotation_matrix_and_orientation_helpers():
    a, b = torch.tensor([0.0, 0.6, 0.8]), torch.tensor([0.0, 0.0, 1.0])
    R = rotation_matrix(a, b)
    assert torch.allclose(R @ a, b, atol=1e-6) and torch.allclose(R @ R.T, torch.eye(3), atol=1e-6)
    assert abs(torch.linalg.det(R).item() - 1.0) < 1e-5
    R2 = rotation_matrix(b, b)  # parallel: identity
    assert torch.allclose(R2, torch.eye(3), atol=1e-6)
    R3 = rotation_matrix(-b, b)  # anti-parallel: a half turn
    assert torch.allclose(R3 @ -b, b, atol=1e-6)
    with pytest.raises(NotImplementedError):
        auto_orient_and_center_poses(torch.eye(4)[None], method="pca")


def test_downscale_folders_and_camera_rescale(tmp_path):
    make_scene(tmp_path, folders=("images", "thermal"))
    parser = ThermalDataParserConfig(data=tmp_path, downscale_factor=2).setup()
    out = parser.get_dataparser_outputs("train")
    assert out.image_filenames[0].parent.name == "images_2" and out.metadata["thermal"][0].parent.name == "thermal_2"
    assert (out.cameras.height, out.cameras.width) == (3, 4) and out.cameras.fx[0].item() == 5.0 and out.cameras.cx == 2.0


def test_split_lists_per_frame_intrinsics_and_refusals(tmp_path):
    names = make_scene(tmp_path / "a", extra={"train_filenames": ["images/frame_train_0001.png", "images/frame_eval_0000.png"],
                                              "val_filenames": ["images/frame_train_0000.png"]})
    p = ThermalDataParserConfig(data=tmp_path / "a").setup()
    assert [f.name for f in p.get_dataparser_outputs("train").image_filenames] == ["frame_eval_0000.png", "frame_train_0001.png"]
    with pytest.raises(RuntimeError, match="missing"):
        p.get_dataparser_outputs("test")
    make_scene(tmp_path / "b", per_frame_intrinsics=True)
    cams = ThermalDataParserConfig(data=tmp_path / "b").setup().get_dataparser_outputs("train").cameras
    assert cams.fx.shape == (5,) and len(set(cams.fx.tolist())) == 5 and cams.width == 8
    make_scene(tmp_path / "c", extra={"k1": 0.1, "p2": -0.01})
    cams = ThermalDataParserConfig(data=tmp_path / "c").setup().get_dataparser_outputs("train").cameras
    assert cams.distortion_params.shape == (5, 6)
    assert torch.allclose(cams.distortion_params[3], torch.tensor([0.1, 0.0, 0.0, 0.0, 0.0, -0.01]))  # k1 k2 k3 k4 p1 p2
    assert torch.equal(ThermalDataParserConfig(data=tmp_path / "b").setup().get_dataparser_outputs("train").cameras
                       .distortion_params, torch.zeros(5, 6))
    make_scene(tmp_path / "d", extra={"camera_model": "OPENCV_FISHEYE"})
    with pytest.raises(NotImplementedError, match="perspective"):
        ThermalDataParserConfig(data=tmp_path / "d").setup().get_dataparser_outputs("train")


def test_dataset_items(tmp_path):
    make_scene(tmp_path)
    out = ThermalDataParserConfig(data=tmp_path).setup().get_dataparser_outputs("train")
    ds = ThermalDataset(out)
    assert len(ds) == 5
    item = ds[2]
    assert item["image_idx"] == 2 and item["image"].shape == (6, 8, 3) and item["thermal"].shape == (6, 8, 1)
    assert item["image"].dtype == torch.float32 and item["thermal"].dtype == torch.float32
    raw = np.array(Image.open(out.metadata["thermal"][2]))
    assert torch.equal(item["thermal"][..., 0], torch.from_numpy((raw / 255.0).astype(np.float32)))
    rgb = np.array(Image.open(out.image_filenames[2]))
    assert torch.equal(item["image"], torch.from_numpy(rgb.astype(np.float32) / 255.0))
    # NS InputDataset deep-copies the cameras: a second dataset on the same outputs does not rescale them again
    fx0 = float(out.cameras.fx.reshape(-1)[0])
    a, b = ThermalDataset(out, scale_factor=0.5), ThermalDataset(out, scale_factor=0.5)
    assert float(out.cameras.fx.reshape(-1)[0]) == fx0
    assert float(a.cameras.fx.reshape(-1)[0]) == float(b.cameras.fx.reshape(-1)[0]) == 0.5 * fx0
    out.metadata.pop("thermal")
    with pytest.raises(AssertionError):
        ThermalDataset(out)  # REF thermal_dataset.py:30


def test_reference_thermal_image_and_temperature_bounds():
    """REF tests/data/thermal/IMG_3561.PNG (640x480 FLIR greyscale) through get_thermal_tensors_from_path."""
    t = ThermalDataset.get_thermal_tensors_from_path(GOLDEN / "IMG_3561.PNG")
    assert t.shape == (640, 480, 1) and t.dtype == torch.float32
    raw = np.array(Image.open(GOLDEN / "IMG_3561.PNG"))
    assert raw.dtype == np.uint8 and raw.ndim == 2
    assert torch.equal(t[..., 0] * 255.0, torch.from_numpy(raw.astype(np.float32)))  # /255 then *255 is exact on uint8 values
    assert 0.0 <= t.min().item() < t.max().item() <= 1.0
    half = ThermalDataset.get_thermal_tensors_from_path(GOLDEN / "IMG_3561.PNG", scale_factor=0.5)
    assert half.shape == (320, 240, 1) and abs(half.mean().item() - t.mean().item()) < 0.01
    # cv2.resize INTER_LINEAR [REF thermal_dataset.py:66-70] at exactly 1/2: source coordinate 2d + 0.5 -> the un-antialiased
    # bilinear tap is the mean of the 2x2 block
    blocks = t[..., 0].view(320, 2, 240, 2)
    want = (blocks[:, 0, :, 0] + blocks[:, 0, :, 1] + blocks[:, 1, :, 0] + blocks[:, 1, :, 1]) / 4
    assert (half[..., 0] - want).abs().max().item() < 1e-6
    third = ThermalDataset.get_thermal_tensors_from_path(GOLDEN / "IMG_3561.PNG", scale_factor=0.3)
    assert third.shape == (192, 144, 1)
    # 0.3: source x = (d + 0.5) / 0.3 - 0.5; check one interior tap by hand (no antialiasing: exactly 4 source pixels)
    sy, sx = (7 + 0.5) * (640 / 192) - 0.5, (11 + 0.5) * (480 / 144) - 0.5
    y0, x0 = int(sy), int(sx)
    fy, fx = sy - y0, sx - x0
    g = t[..., 0]
    hand = ((1 - fy) * ((1 - fx) * g[y0, x0] + fx * g[y0, x0 + 1]) + fy * ((1 - fx) * g[y0 + 1, x0] + fx * g[y0 + 1, x0 + 1]))
    assert abs(third[7, 11, 0].item() - hand.item()) < 1e-5
    bounds = json.loads((GOLDEN / "temperature_bounds.json").read_text())
    lo, hi = bounds["absolute_min_temperature"], bounds["absolute_max_temperature"]
    celsius = t * (hi - lo) + lo  # the mapping mae_thermal applies [REF thermal_metrics.py:29-33]
    assert lo - 1e-4 <= celsius.min().item() and celsius.max().item() <= hi + 1e-4  # fp32 rounding of the affine map
    with pytest.raises(FileNotFoundError):
        ThermalDataset.get_thermal_tensors_from_path(GOLDEN / "missing.png")
