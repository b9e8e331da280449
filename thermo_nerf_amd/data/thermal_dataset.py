"""``ThermalDataset`` [REF thermo_nerf/thermal_nerf/thermal_dataset.py:13-73] (+ the RGB side of nerfstudio's InputDataset):
item i = {"image_idx", "image" [H,W,3] float32 in [0,1], "thermal" [H,W,1] float32 in [0,1]}."""
from __future__ import annotations

import copy
from pathlib import Path
from typing import Dict, List

import numpy as np
import torch
from PIL import Image

from ..rendered_image_modalities import RenderedImageModality
from .thermal_dataparser import DataparserOutputs


class ThermalDataset:
    def __init__(self, dataparser_outputs: DataparserOutputs, scale_factor: float = 1.0, kernel_size: int = 3) -> None:
        self._dataparser_outputs = dataparser_outputs
        self.scale_factor = scale_factor
        self.metadata = dataparser_outputs.metadata
        assert RenderedImageModality.THERMAL.value in self.metadata.keys()  # REF :30
        self.thermal_filenames: List[Path] = self.metadata[RenderedImageModality.THERMAL.value]
        self.kernel_size = kernel_size
        # NS InputDataset deep-copies the cameras: two datasets built from one DataparserOutputs must not rescale twice
        self.cameras = copy.deepcopy(dataparser_outputs.cameras)
        self.scene_box = dataparser_outputs.scene_box
        if scale_factor != 1.0:
            self.cameras.rescale_output_resolution(scale_factor)

    def __len__(self) -> int:
        return len(self._dataparser_outputs.image_filenames)

    @property
    def image_filenames(self) -> List[Path]:
        return self._dataparser_outputs.image_filenames

    def get_image(self, image_idx: int) -> torch.Tensor:
        """NS InputDataset.get_image_float32: PIL -> uint8 -> /255; an alpha channel is blended onto black
        (background "last_sample" carries no colour, see RGBRenderer.blend_background_for_loss_computation)."""
        pil = Image.open(self.image_filenames[image_idx])
        if self.scale_factor != 1.0:
            w, h = pil.size
            pil = pil.resize((int(w * self.scale_factor), int(h * self.scale_factor)), resample=Image.Resampling.BILINEAR)
        arr = np.array(pil, dtype="uint8")
        if arr.ndim == 2:
            arr = arr[:, :, None].repeat(3, axis=2)
        assert arr.ndim == 3 and arr.shape[2] in (3, 4), f"Image shape of {arr.shape} is incorrect."
        img = torch.from_numpy(arr.astype("float32") / 255.0)
        if img.shape[-1] == 4:
            img = img[:, :, :3] * img[:, :, -1:]
        return img

    @staticmethod
    def get_thermal_tensors_from_path(filepath: Path, scale_factor: float = 1.0) -> torch.Tensor:
        """[REF :50-73]: greyscale read, /255, float32, optional resize, [H,W,1].  (The reference reads with
        cv2.IMREAD_GRAYSCALE; ThermoScenes thermal PNGs are single-channel, where PIL yields the same bytes.)"""
        filepath = Path(filepath)
        if not filepath.exists():
            raise FileNotFoundError(f"No file found at {filepath}")
        pil = Image.open(filepath)
        if pil.mode != "L":
            pil = pil.convert("L")
        image = (np.asarray(pil, dtype=np.uint8) / 255.0).astype(np.float32)
        if scale_factor != 1.0:
            h, w = image.shape
            # cv2.resize's default INTER_LINEAR [REF :66-70] is NOT antialiased (PIL's BILINEAR is, when shrinking): plain
            # bilinear taps at half-pixel centres = torch's bilinear interpolation with align_corners=False
            size = (int(h * scale_factor), int(w * scale_factor))
            image = torch.nn.functional.interpolate(torch.from_numpy(image)[None, None], size=size, mode="bilinear",
                                                    align_corners=False, antialias=False)[0, 0].numpy()
        return torch.from_numpy(np.ascontiguousarray(image[:, :, np.newaxis]))

    def get_metadata(self, data: Dict) -> Dict[str, torch.Tensor]:
        """[REF :35-48]"""
        filepath = Path(self.thermal_filenames[data["image_idx"]])
        return {RenderedImageModality.THERMAL.value: self.get_thermal_tensors_from_path(filepath, self.scale_factor)}

    def get_data(self, image_idx: int) -> Dict:
        data = {"image_idx": image_idx, "image": self.get_image(image_idx)}
        data.update(self.get_metadata(data))
        return data

    def __getitem__(self, image_idx: int) -> Dict:
        return self.get_data(image_idx)

    def to_ray_table(self, device="cuda"):
        """Every pixel of every image as one HBM-resident table for ``thermo_nerf_amd.trainer.Trainer``."""
        from ..trainer import RayDataset

        items = [self[i] for i in range(len(self))]
        return RayDataset.from_images(self.cameras, [it["image"] for it in items],
                                      [it[RenderedImageModality.THERMAL.value] for it in items], device)
