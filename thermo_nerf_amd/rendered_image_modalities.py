"""Keys of the model's output dictionary / of the images an evaluation saves.  The STRING VALUES are part of the drop-in
boundary (they index ``outputs[...]``, ``batch[...]`` and name saved files) and are therefore the reference's own
[REF thermo_nerf/rendered_image_modalities.py:4-9]; everything else about this module is free."""
from enum import Enum

# member name -> dictionary key; "img" is nerfstudio's key for the (ground truth | prediction) RGB strip
_KEYS = {
    "RGB": "img",
    "DEPTH": "depth",
    "ACCUMULATION": "accumulation",
    "THERMAL": "thermal",
    "THERMAL_COMBINED": "thermal_combined",
}

RenderedImageModality = Enum("RenderedImageModality", _KEYS, module=__name__)
