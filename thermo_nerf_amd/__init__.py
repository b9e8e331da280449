"""thermo_nerf_amd — MI355X (gfx950) implementation of ThermoNeRF's volumetric-rendering hot path.

The package mirrors the slice of the reference's plugin surface that sits on that path
(ThermalNerfModel.get_outputs, ThermalNerfactoTField.get_density/get_outputs, ThermalRenderer, the
nerfstudio samplers/renderers they call) and routes every arithmetic step to hand-written HIP kernels in
``libthermonerf_hip.so`` through the C-ABI of ``include/thermonerf_hip.h``.  There is no CPU or PyTorch
fallback: without the shared object, or on a CPU tensor, calls raise.
"""
from .rendered_image_modalities import RenderedImageModality  # noqa: F401
from .rays import Frustums, RayBundle, RaySamples  # noqa: F401
from .scene import NearFarCollider, SceneBox, SceneContraction  # noqa: F401
from .fields import FieldHeadNames, HashMLPDensityField  # noqa: F401
from .thermal_nerf.thermal_field_head import FieldHeadNamesT  # noqa: F401
from .thermal_nerf.thermal_field import ThermalNerfactoTField  # noqa: F401
from .thermal_nerf.thermal_renderer import ThermalRenderer  # noqa: F401
from .thermal_nerf.thermal_nerf_model import ThermalNerfModel, ThermalNerfModelConfig  # noqa: F401

__version__ = "0.1.0"
