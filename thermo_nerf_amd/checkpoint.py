"""nerfstudio checkpoint import/export for ``ThermalNerfModel``.

The reference's harness takes the LAST ``*.ckpt`` under the run directory, ``torch.load``s it and hands
``loaded_state["pipeline"]`` / ``loaded_state["step"]`` to ``Pipeline.load_pipeline``
[REF thermo_nerf/render/renderer.py:94-113].  The pipeline state dict prefixes every model parameter with ``_model.``
(``module.`` on top under DDP, stripped by nerfstudio) and also carries keys this path does not own (``datamanager.*``,
the LPIPS network of the metrics).  This module reads exactly that layout — tensors only (``weights_only=True``; the
reference's ``config.yml`` is a python-object YAML and is never touched) — into the modules whose names already are
nerfstudio's, and writes the same layout back so a model can go the other way.

No reference-trained checkpoint exists in this checkout (the test blob is missing, REF .MISSING_LARGE_BLOBS:4), so the
key names are pinned by nerfstudio 1.1.5's module structure and by REF tests/test_renderer.py:33-37
(``…mlp_base.layers.N.weight`` naming), not by a real file.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Mapping, Optional, Tuple, Union

import torch
from torch import Tensor

MODEL_PREFIX = "_model."
# pipeline entries that belong to other subsystems (kept out of the hot path on purpose)
FOREIGN_PREFIXES = ("datamanager.", "lpips.", "psnr.", "ssim.", "_model.lpips.", "_model.psnr.", "_model.ssim.")
# tiny-cuda-nn packs encoder + MLP of a module into one flat ``params`` vector whose grid indexing and hidden padding
# differ from the torch modules: such a checkpoint cannot be mapped onto torch-layout tables.
_TCNN_KEY = re.compile(r"\.(tcnn_encoding|model)\.params$")
# MLPWithHashEncoding exposes (encoder, mlp); some nerfstudio builds alias them as a Sequential ``model``
_ALIASES = ((re.compile(r"\.mlp_base\.model\.0\."), ".mlp_base.encoder."), (re.compile(r"\.mlp_base\.model\.1\."), ".mlp_base.mlp."))


@dataclass
class LoadReport:
    step: Optional[int]
    loaded: List[str] = field(default_factory=list)
    ignored: List[str] = field(default_factory=list)       # foreign subsystems
    missing: List[str] = field(default_factory=list)       # model entries the checkpoint lacks (non-strict only)
    unexpected: List[str] = field(default_factory=list)    # `_model.` entries the model has no slot for (non-strict only)


def latest_checkpoint(run_dir: Union[str, os.PathLike]) -> Path:
    """The reference picks ``list(model_path.rglob("*.ckpt"))[-1]`` [REF renderer.py:94]; rglob order is filesystem
    order, so sort by the step encoded in nerfstudio's ``step-%09d.ckpt`` (falling back to the name)."""
    paths = list(Path(run_dir).rglob("*.ckpt"))
    if not paths:
        raise FileNotFoundError(f"no *.ckpt under {run_dir}")

    def key(p: Path) -> Tuple[int, str]:
        m = re.search(r"step-(\d+)", p.name)
        return (int(m.group(1)) if m else -1, p.name)

    return sorted(paths, key=key)[-1]


def model_state_from_pipeline(pipeline_state: Mapping[str, Tensor]) -> Tuple[Dict[str, Tensor], List[str]]:
    """``loaded_state["pipeline"]`` -> (model state dict with nerfstudio's prefixes removed, ignored foreign keys)."""
    out: Dict[str, Tensor] = {}
    ignored: List[str] = []
    has_prefix = any(k.startswith(MODEL_PREFIX) or k.startswith("module." + MODEL_PREFIX) for k in pipeline_state)
    for k, v in pipeline_state.items():
        if k.startswith("module."):  # DDP wrapper, stripped by Pipeline.load_pipeline
            k = k[len("module."):]
        if any(k.startswith(p) for p in FOREIGN_PREFIXES):
            ignored.append(k)
            continue
        if has_prefix:
            if not k.startswith(MODEL_PREFIX):
                ignored.append(k)
                continue
            k = k[len(MODEL_PREFIX):]
        if _TCNN_KEY.search("." + k):
            raise NotImplementedError(
                f"checkpoint entry '{k}' is a tiny-cuda-nn packed parameter vector; its grid indexing and layer padding "
                "differ from nerfstudio's torch modules, so it cannot be mapped onto this path. Re-train/export with "
                "implementation='torch'.")
        for pat, rep in _ALIASES:
            k = pat.sub(rep, "." + k)[1:] if pat.search("." + k) else k
        out[k] = v
    return out, ignored


def load_nerfstudio_checkpoint(model: torch.nn.Module, source: Union[str, os.PathLike, Mapping], strict: bool = True
                               ) -> LoadReport:
    """Load a nerfstudio ``step-*.ckpt`` (path, run directory, or the already-loaded dict) into ``model``.

    ``strict`` (default) demands that every parameter of the model is present with the right shape and that no
    ``_model.`` entry is left over, like ``Pipeline.load_pipeline``'s ``load_state_dict``.  Shape mismatches always
    raise, naming the usual cause (``num_train_data`` / table sizes differ from the training run).
    """
    if isinstance(source, Mapping):
        state = source
    else:
        path = Path(source)
        if path.is_dir():
            path = latest_checkpoint(path)
        state = torch.load(path, map_location="cpu", weights_only=True)
    step = None
    if "pipeline" in state:
        step = int(state["step"]) if "step" in state else None
        state = state["pipeline"]
    incoming, ignored = model_state_from_pipeline(state)
    own = model.state_dict()
    report = LoadReport(step=step, ignored=ignored)
    bad_shapes = []
    for k, v in incoming.items():
        if k not in own:
            report.unexpected.append(k)
        elif tuple(own[k].shape) != tuple(v.shape):
            bad_shapes.append(f"{k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
        else:
            report.loaded.append(k)
    report.missing = [k for k in own if k not in incoming]
    if bad_shapes:
        raise ValueError("checkpoint does not fit the model (build the model with the training run's num_train_data, "
                         "hash-table and sample-count settings): " + "; ".join(bad_shapes))
    # buffers are constants derived from the config (aabb, scalings, max_res ...): whether nerfstudio registers each of
    # them differs between releases, so only PARAMETERS are mandatory
    params = {k for k, _ in model.named_parameters()}
    missing_params = [k for k in report.missing if k in params]
    if strict and (missing_params or report.unexpected):
        raise KeyError(f"strict checkpoint load failed: missing {missing_params}, unexpected {report.unexpected}")
    with torch.no_grad():
        for k in report.loaded:
            own[k].copy_(incoming[k].to(device=own[k].device, dtype=own[k].dtype))  # bumps _version: blobs re-prepared
    return report


def save_nerfstudio_checkpoint(model: torch.nn.Module, directory: Union[str, os.PathLike], step: int) -> Path:
    """Write ``<directory>/step-%09d.ckpt`` with the keys the reference reads (``step``, ``pipeline``)
    [REF renderer.py:94-113]; optimizer/scheduler state is the trainer's business and is not written."""
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    path = directory / f"step-{step:09d}.ckpt"
    pipeline = {MODEL_PREFIX + k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({"step": int(step), "pipeline": pipeline}, path)
    return path
