"""Output-dict keys of the model [REF thermo_nerf/rendered_image_modalities.py:4-9] — part of the boundary."""
from enum import Enum


class RenderedImageModality(Enum):
    RGB = "img"
    DEPTH = "depth"
    ACCUMULATION = "accumulation"
    THERMAL = "thermal"
    THERMAL_COMBINED = "thermal_combined"
