"""Evaluation harness — counterpart of the reference's ``Evaluator`` [REF thermo_nerf/evaluator/evaluator.py:15-160]:
render every image of the eval split ONCE (all modalities come out of one pass), compute the per-image metrics of
``ThermalNerfModel.get_image_metrics_and_images`` (RGB / thermal PSNR and SSIM, thermal MAE in degrees for the whole image
and for the foreground beyond ``threshold``; LPIPS is NaN — its pretrained network does not exist offline), aggregate
``<key>``, ``<key>_mean``, ``<key>_std`` exactly as the reference does, write ``metrics.json`` and the rendered images.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from PIL import Image

from .rays import RayBundle
from .rendered_image_modalities import RenderedImageModality


class Evaluator:
    def __init__(self, model, eval_dataset, experiment_name: str = "", method_name: str = "thermal-nerf",
                 job_param_identifier: Optional[str] = None,
                 modalities_to_save: Sequence[RenderedImageModality] = (RenderedImageModality.RGB,),
                 threshold: Optional[float] = None, device="cuda") -> None:
        self.model, self.dataset, self.device = model, eval_dataset, device
        self.identifier = job_param_identifier
        self.modalities_to_save = list(modalities_to_save)
        self._evaluation_images: Dict[RenderedImageModality, List[np.ndarray]] = {m: [] for m in self.modalities_to_save}
        self._metrics = self._compute_metrics(threshold)
        self._benchmark_info = {"experiment_name": experiment_name, "method_name": method_name,
                                "job_param_identifier": self.identifier, "results": self._metrics}

    @property
    def metrics(self) -> Dict:
        return self._metrics

    @torch.no_grad()
    def _compute_metrics(self, threshold: Optional[float]) -> Dict:
        """[REF evaluator.py:47-106]"""
        model = self.model
        was_training = model.training
        model.eval()
        per_image: List[Dict[str, float]] = []
        cams = self.dataset.cameras
        for idx in range(len(self.dataset)):
            item = self.dataset[idx]
            rb = cams.generate_rays(idx, device=self.device)
            h, w = rb.origins.shape[:2]
            flat = rb.flatten()
            # REF :68-76: the eval loader hands ONE camera at a time and the reference builds camera_indices =
            # arange(cameras.camera_to_worlds.shape[0]) = [0] for it, so every eval image is adjusted with row 0 of the
            # optimizer's table (never with the eval-split index, which may exceed num_train_data)
            if flat.camera_indices is not None:
                flat.camera_indices = torch.zeros_like(flat.camera_indices)
            model.camera_optimizer.apply_to_raybundle(flat)
            rb = RayBundle(origins=flat.origins.view(h, w, 3), directions=flat.directions.view(h, w, 3),
                           pixel_area=rb.pixel_area, camera_indices=rb.camera_indices)
            outputs = model.get_outputs_for_camera_ray_bundle(rb)
            batch = {"image": item["image"].to(self.device),
                     RenderedImageModality.THERMAL.value: item[RenderedImageModality.THERMAL.value].to(self.device)}
            # REF :82-87: metrics and images of the frame come from the model
            metrics, images = model.get_image_metrics_and_images(outputs, batch, threshold=threshold)
            per_image.append(metrics)
            for m in self.modalities_to_save:  # REF :89-92
                self._evaluation_images[m].append((images[m.value].clamp(0, 1) * 255).byte().cpu().numpy())
        model.train(was_training)
        if not per_image:
            raise RuntimeError("Cannot evaluate without eval images")
        out: Dict = {}
        for key in per_image[0]:
            vals = torch.tensor([m[key] for m in per_image], dtype=torch.float64)
            std, mean = torch.std_mean(vals) if len(per_image) > 1 else (torch.tensor(float("nan")), vals.mean())
            out[f"{key}_mean"], out[f"{key}_std"] = float(mean), float(std)
            out[key] = [m[key] for m in per_image]
        return out

    def save_images(self, modalities: Sequence[RenderedImageModality], output_path) -> None:
        """[REF evaluator.py:108-124]"""
        output_path = Path(output_path)
        output_path.mkdir(parents=True, exist_ok=True)
        for m in modalities:
            for idx, image in enumerate(self._evaluation_images[m]):
                arr = image[:, :, 0] if image.shape[-1] == 1 else image
                Image.fromarray(arr).save(output_path / f"{m.value}_{idx:05d}.jpg")

    def save_metrics(self, output_folder) -> Path:
        """[REF evaluator.py:126-133]"""
        output_file = Path(output_folder, "metrics.json")
        output_file.parent.mkdir(parents=True, exist_ok=True)
        def strict(v):  # NaN (LPIPS without its pretrained weights, the std of a single image) is not JSON: write null
            if isinstance(v, float) and v != v:
                return None
            if isinstance(v, dict):
                return {k: strict(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return [strict(x) for x in v]
            return v

        output_file.write_text(json.dumps(strict(self._benchmark_info), indent=2, allow_nan=False), "utf8")
        return output_file
