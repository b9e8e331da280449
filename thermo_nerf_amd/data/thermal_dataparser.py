"""``Thermal`` dataparser [REF thermo_nerf/thermal_nerf/thermal_dataparser.py:30-343]: nerfstudio ``transforms.json`` with a
per-frame ``thermal_file_path`` -> image / thermal file lists, cameras, scene box, the pose normalisation.

Restates the ``Nerfstudio`` dataparser pieces the reference inherits (``_get_fname`` with its ``images_<k>`` /
``thermal_<k>`` downscale folders, the filename split ``frame_train_* / frame_eval_*`` [REF thermo_scenes/docs/
Collect_new_dataset.md:105-130], auto orient / centre / scale).  Limits, raised explicitly: perspective (OPENCV / pinhole)
cameras with one shared principal point and image size."""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from ..cameras import Cameras
from ..rendered_image_modalities import RenderedImageModality
from ..scene import SceneBox
from .camera_utils import auto_orient_and_center_poses


@dataclass
class ThermalDataParserConfig:
    """[REF thermal_dataparser.py:30-52] + the NerfstudioDataParserConfig defaults it inherits."""

    data: Path = Path("data/DTU/scan65")
    scale_factor: float = 1.0
    downscale_factor: Optional[int] = 1
    scene_scale: float = 1.0
    orientation_method: str = "up"
    center_method: str = "poses"
    auto_scale_poses: bool = True
    eval_mode: str = "filename"
    train_split_fraction: float = 0.9
    eval_interval: int = 8
    depth_unit_scale_factor: float = 1e-3

    def setup(self) -> "Thermal":
        return Thermal(self)


@dataclass
class DataparserOutputs:
    """NS DataparserOutputs (the fields the reference fills, REF :326-342)."""

    image_filenames: List[Path]
    cameras: Cameras
    scene_box: SceneBox
    dataparser_scale: float
    dataparser_transform: Tensor
    metadata: Dict = field(default_factory=dict)
    mask_filenames: Optional[List[Path]] = None


def get_train_eval_split_filename(image_filenames: List[Path]) -> Tuple[np.ndarray, np.ndarray]:
    """NS dataparsers_utils.get_train_eval_split_filename: "train" / "eval" in the file's base name."""
    basenames = [Path(f).name for f in image_filenames]
    i_all = np.arange(len(image_filenames))
    i_train = np.array([i for i, b in zip(i_all, basenames) if "train" in b], dtype=np.int64)
    i_eval = np.array([i for i, b in zip(i_all, basenames) if "eval" in b], dtype=np.int64)
    return i_train, i_eval


def get_train_eval_split_fraction(image_filenames: List[Path], train_split_fraction: float) -> Tuple[np.ndarray, np.ndarray]:
    """NS get_train_eval_split_fraction: evenly spaced training images, the rest evaluates."""
    n = len(image_filenames)
    n_train = math_ceil(n * train_split_fraction)
    i_all = np.arange(n)
    i_train = np.linspace(0, n - 1, n_train, dtype=int)
    return i_train, np.setdiff1d(i_all, i_train)


def math_ceil(x: float) -> int:
    return int(np.ceil(x))


def get_train_eval_split_interval(image_filenames: List[Path], eval_interval: int) -> Tuple[np.ndarray, np.ndarray]:
    i_all = np.arange(len(image_filenames))
    return i_all[i_all % eval_interval != 0], i_all[i_all % eval_interval == 0]


class Thermal:
    def __init__(self, config: ThermalDataParserConfig) -> None:
        self.config = config
        self.downscale_factor: Optional[int] = config.downscale_factor

    def get_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        return self._generate_dataparser_outputs(split)

    def _get_fname(self, filepath: Path, data_dir: Path, downsample_folder_prefix: str = "images_") -> Path:
        """NS Nerfstudio._get_fname: with downscale_factor k > 1 the file lives in ``<prefix><k>/`` next to the original."""
        if self.downscale_factor is None:
            raise NotImplementedError("automatic downscale_factor selection needs the image sizes; set downscale_factor")
        if self.downscale_factor > 1:
            return data_dir / f"{downsample_folder_prefix}{self.downscale_factor}" / filepath.name
        return data_dir / filepath

    def _generate_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        cfg = self.config
        data = Path(cfg.data)
        if data.suffix == ".json":
            meta, data_dir = json.loads(data.read_text()), data.parent
        else:
            meta, data_dir = json.loads((data / "transforms.json").read_text()), data

        if meta.get("camera_model", "OPENCV") not in ("OPENCV", "PINHOLE", "SIMPLE_PINHOLE"):
            raise NotImplementedError(f"camera_model {meta['camera_model']}: only perspective cameras")

        frames = sorted(meta["frames"], key=lambda f: str(self._get_fname(Path(f["file_path"]), data_dir)))  # REF :100-107
        dist_keys = ("k1", "k2", "k3", "k4", "p1", "p2")  # NS camera_utils.get_distortion_params order
        distort_fixed = any(k in meta for k in ("k1", "k2", "k3", "p1", "p2"))  # REF :89-93
        distort: List[List[float]] = []
        image_filenames, thermal_filenames, poses = [], [], []
        intr = {k: [] for k in ("fl_x", "fl_y", "cx", "cy", "h", "w")}
        for frame in frames:
            image_filenames.append(self._get_fname(Path(frame["file_path"]), data_dir))
            poses.append(np.array(frame["transform_matrix"]))
            for k in intr:
                if k not in meta:
                    assert k in frame, f"{k} not specified in frame"  # REF :113-130
                    intr[k].append(float(frame[k]))
            if not distort_fixed:  # REF :131-141
                distort.append([float(frame[k]) if k in frame else 0.0 for k in dist_keys])
            if "thermal_file_path" in frame:  # REF :146-153
                thermal_filenames.append(self._get_fname(Path(frame["thermal_file_path"]), data_dir,
                                                         downsample_folder_prefix="thermal_"))

        has_split_files_spec = any(f"{s}_filenames" in meta for s in ("train", "val", "test"))
        if f"{split}_filenames" in meta:  # REF :158-176
            split_filenames = set(self._get_fname(Path(x), data_dir) for x in meta[f"{split}_filenames"])
            unmatched = split_filenames.difference(image_filenames)
            if unmatched:
                raise RuntimeError(f"Some filenames for split {split} were not found: {unmatched}.")
            indices = np.array([i for i, p in enumerate(image_filenames) if p in split_filenames], dtype=np.int64)
        elif has_split_files_spec:
            raise RuntimeError(f"The dataset's list of filenames for split {split} is missing.")
        else:  # REF :181-204
            if cfg.eval_mode == "fraction":
                i_train, i_eval = get_train_eval_split_fraction(image_filenames, cfg.train_split_fraction)
            elif cfg.eval_mode == "filename":
                i_train, i_eval = get_train_eval_split_filename(image_filenames)
            elif cfg.eval_mode == "interval":
                i_train, i_eval = get_train_eval_split_interval(image_filenames, cfg.eval_interval)
            elif cfg.eval_mode == "all":
                i_train = i_eval = np.arange(len(image_filenames))
            else:
                raise ValueError(f"Unknown eval mode {cfg.eval_mode}")
            if split == "train":
                indices = i_train
            elif split in ("val", "test"):
                indices = i_eval
            else:
                raise ValueError(f"Unknown dataparser split {split}")

        orientation = meta.get("orientation_override", cfg.orientation_method)
        poses_t = torch.from_numpy(np.array(poses).astype(np.float32))
        poses_t, transform_matrix = auto_orient_and_center_poses(poses_t, method=orientation, center_method=cfg.center_method)
        scale_factor = 1.0
        if cfg.auto_scale_poses:  # REF :215-219
            scale_factor /= float(torch.max(torch.abs(poses_t[:, :3, 3])))
        scale_factor *= cfg.scale_factor
        poses_t[:, :3, 3] *= scale_factor

        idx = torch.as_tensor(np.asarray(indices), dtype=torch.long)
        image_filenames = [image_filenames[i] for i in indices]
        thermal_filenames = [thermal_filenames[i] for i in indices] if thermal_filenames else []
        poses_t = poses_t[idx]

        def shared(key: str) -> float:
            if key in meta:
                return float(meta[key])
            vals = set(intr[key])
            if len(vals) != 1:
                raise NotImplementedError(f"per-frame {key} differs between frames; one shared value is supported")
            return vals.pop()

        n = len(indices)
        fx = torch.full((n,), float(meta["fl_x"])) if "fl_x" in meta else torch.tensor(intr["fl_x"], dtype=torch.float32)[idx]
        fy = torch.full((n,), float(meta["fl_y"])) if "fl_y" in meta else torch.tensor(intr["fl_y"], dtype=torch.float32)[idx]
        if distort_fixed:  # REF :287-296
            distortion = torch.tensor([float(meta[k]) if k in meta else 0.0 for k in dist_keys]).expand(n, 6).contiguous()
        else:
            distortion = torch.tensor(distort, dtype=torch.float32).reshape(-1, 6)[idx]
        cameras = Cameras(camera_to_worlds=poses_t[:, :3, :4].contiguous(), fx=fx, fy=fy, cx=shared("cx"), cy=shared("cy"),
                          height=int(shared("h")), width=int(shared("w")), distortion_params=distortion)
        cameras.rescale_output_resolution(1.0 / self.downscale_factor)  # REF :305-306

        if "applied_transform" in meta:  # REF :308-319
            applied = torch.tensor(meta["applied_transform"], dtype=transform_matrix.dtype)
            transform_matrix = transform_matrix @ torch.cat([applied, torch.tensor([[0, 0, 0, 1]], dtype=transform_matrix.dtype)], 0)
        if "applied_scale" in meta:
            scale_factor *= float(meta["applied_scale"])

        s = cfg.scene_scale
        scene_box = SceneBox(aabb=torch.tensor([[-s, -s, -s], [s, s, s]], dtype=torch.float32))
        return DataparserOutputs(
            image_filenames=image_filenames, cameras=cameras, scene_box=scene_box, dataparser_scale=scale_factor,
            dataparser_transform=transform_matrix,
            metadata={RenderedImageModality.THERMAL.value: thermal_filenames if thermal_filenames else None})
