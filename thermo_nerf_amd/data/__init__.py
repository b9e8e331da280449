"""Dataset boundary of the hot path: the reference's ``Thermal`` dataparser and ``ThermalDataset``
restated without nerfstudio, feeding the HBM-resident ray table of ``thermo_nerf_amd.trainer``.  Pure host code."""
from .thermal_dataparser import DataparserOutputs, Thermal, ThermalDataParserConfig  # noqa: F401
from .thermal_dataset import ThermalDataset  # noqa: F401
