"""Perspective cameras, camera-path loading and per-frame evaluation — the callers on either side of the hot path
(SURVEY §8f-1): nerfstudio ``Cameras.generate_rays`` / ``get_path_from_json`` as used by the reference's harnesses
[REF thermo_nerf/render/renderer.py:144-201; thermo_nerf/evaluator/evaluator.py:47-106].

Rays are generated ON the device by ``tn_generate_rays`` (nothing crosses PCIe per frame), a frame is rendered ONCE
for all modalities (the reference loops modality-outer and renders every frame per modality, REF renderer.py:180-183).
"""
from __future__ import annotations

import ctypes
import json
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _hip
from .rays import RayBundle
from .thermal_nerf.thermal_metrics import mae_thermal


@dataclass
class Cameras:
    """NS Cameras, perspective only: camera_to_worlds [N,3,4]; fx, fy [N]; cx, cy scalars; height, width ints."""

    camera_to_worlds: Tensor
    fx: Tensor
    fy: Tensor
    cx: float
    cy: float
    height: int
    width: int
    distortion_params: Optional[Tensor] = None  # [N,6] (k1, k2, k3, k4, p1, p2) as NS get_distortion_params, or None

    def __len__(self) -> int:
        return self.camera_to_worlds.shape[0]

    @property
    def size(self) -> int:
        return len(self)

    def rescale_output_resolution(self, scaling_factor: float) -> None:
        """NS Cameras.rescale_output_resolution (used by REF renderer.py:157)."""
        self.fx = self.fx * scaling_factor
        self.fy = self.fy * scaling_factor
        self.cx = self.cx * scaling_factor
        self.cy = self.cy * scaling_factor
        self.height = int(self.height * scaling_factor)
        self.width = int(self.width * scaling_factor)

    def generate_rays(self, camera_indices: int, device="cuda", rows: Optional[Tuple[int, int]] = None,
                      flat: bool = False) -> RayBundle:
        """NS Cameras.generate_rays(camera_indices=i): an [H,W] RayBundle (or rows [r0,r1) of it), on the device."""
        idx = int(camera_indices)
        r0, r1 = rows if rows is not None else (0, self.height)
        n = (r1 - r0) * self.width
        dev = torch.device(device)
        o = torch.empty((n, 3), dtype=torch.float32, device=dev)
        d = torch.empty((n, 3), dtype=torch.float32, device=dev)
        area = torch.empty((n, 1), dtype=torch.float32, device=dev)
        _hip.require_device_tensor(o, "origins")
        c2w = (ctypes.c_float * 12)(*self.camera_to_worlds[idx].reshape(-1).tolist())
        dist = None
        if self.distortion_params is not None:
            dist = (ctypes.c_float * 6)(*self.distortion_params[idx].reshape(-1).tolist())
        lib = _hip.load()
        with torch.cuda.device(dev):
            _hip.check(
                lib.tn_generate_rays(c2w, float(self.fx[idx]), float(self.fy[idx]), float(self.cx), float(self.cy),
                                     self.height, self.width, dist, r0 * self.width, n, o.data_ptr(), d.data_ptr(),
                                     area.data_ptr(), _hip.current_stream()),
                "tn_generate_rays",
            )
        cam = torch.full((n, 1), idx, dtype=torch.long, device=dev)
        if flat:
            return RayBundle(origins=o, directions=d, pixel_area=area, camera_indices=cam)
        shape = (r1 - r0, self.width)
        return RayBundle(origins=o.view(*shape, 3), directions=d.view(*shape, 3), pixel_area=area.view(*shape, 1),
                         camera_indices=cam.view(*shape, 1))


def three_js_perspective_camera_focal_length(fov: float, image_height: int) -> float:
    """NS camera_utils helper: focal length in pixels from a vertical field of view in degrees."""
    return (image_height / 2.0) / math.tan(fov * (math.pi / 180.0) / 2.0)


def get_path_from_json(camera_path: Dict) -> Cameras:
    """NS camera_paths.get_path_from_json for perspective paths (what REF renderer.py:154-156 loads)."""
    if camera_path.get("camera_type", "perspective") != "perspective":
        raise NotImplementedError("only perspective camera paths are implemented")
    height, width = int(camera_path["render_height"]), int(camera_path["render_width"])
    c2ws, focal = [], []
    for cam in camera_path["camera_path"]:
        c2ws.append(torch.tensor(cam["camera_to_world"], dtype=torch.float32).view(4, 4)[:3])
        focal.append(three_js_perspective_camera_focal_length(cam["fov"], height))
    f = torch.tensor(focal, dtype=torch.float32)
    return Cameras(camera_to_worlds=torch.stack(c2ws, dim=0), fx=f, fy=f.clone(), cx=width / 2, cy=height / 2,
                   height=height, width=width)


def load_cameras(path, rendered_resolution_scaling_factor: float = 1.0) -> Cameras:
    """Counterpart of Renderer.load_cameras [REF thermo_nerf/render/renderer.py:144-158]."""
    with open(path, "r", encoding="utf-8") as f:
        cameras = get_path_from_json(json.load(f))
    cameras.rescale_output_resolution(rendered_resolution_scaling_factor)
    return cameras


def psnr(pred: Tensor, gt: Tensor) -> Tensor:
    """torchmetrics PeakSignalNoiseRatio(data_range=1.0) [REF thermal_nerf_model.py:200]: 10 log10(1 / MSE)."""
    return 10.0 * torch.log10(1.0 / torch.mean((pred - gt) ** 2))


def frame_metrics(outputs: Dict[str, Tensor], gt_rgb: Tensor, gt_thermal: Tensor, max_temperature: float,
                  min_temperature: float, cold: bool = False, threshold: Optional[float] = None) -> Dict[str, float]:
    """The BASELINE quality metrics of one rendered frame, as get_image_metrics_and_images computes them
    [REF thermal_nerf_model.py:362-391; nerfacto_config/thermal_nerfacto.py:47-84]: RGB PSNR, thermal PSNR and the
    thermal MAE in degrees (whole image and foreground)."""
    rgb, th = outputs["rgb"], outputs["thermal"]
    return {
        "psnr": float(psnr(gt_rgb.to(rgb), rgb)),
        "psnr_thermal": float(psnr(gt_thermal.to(th), th)),
        "mae_thermal": float(mae_thermal(gt_thermal.to(th), th, cold, max_temperature, min_temperature, None)),
        "mae_thermal_foreground": float(mae_thermal(gt_thermal.to(th), th, cold, max_temperature, min_temperature, threshold)),
    }
