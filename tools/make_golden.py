"""Generate the golden fixtures under tests/golden/ (run ONLY in the build container, where /root/reference exists).

G1  thermal_renderer.npz  — inputs + outputs of the REAL reference ``ThermalRenderer`` (train and eval mode),
                            imported from /root/reference/thermo_nerf/thermal_nerf/thermal_renderer.py with
                            annotation-only stubs for ``jaxtyping`` and ``nerfstudio.utils.colors``.
G2  mae_thermal.npz       — inputs + outputs of the REAL reference ``mae_thermal``
                            (/root/reference/thermo_nerf/thermal_nerf/thermal_metrics.py).
G3  oracle_regression.npz — small self-consistency vectors from the CPU oracle (NOT reference-pinned): scalings,
                            hash indices, hash-encode, contraction, piecewise bins, PDF resample.

The fixtures are data (inputs and expected outputs); no reference source text is stored.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _stub_modules() -> None:
    class _Sub:
        def __class_getitem__(cls, item):
            return cls

    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Int", "Shaped"):
        setattr(jt, n, type(n, (_Sub,), {}))
    sys.modules["jaxtyping"] = jt
    ns = types.ModuleType("nerfstudio")
    nsu = types.ModuleType("nerfstudio.utils")
    nsc = types.ModuleType("nerfstudio.utils.colors")
    nsc.COLORS_DICT = {"white": torch.ones(3), "black": torch.zeros(3)}
    nsu.colors = nsc
    ns.utils = nsu
    sys.modules.update({"nerfstudio": ns, "nerfstudio.utils": nsu, "nerfstudio.utils.colors": nsc})


def _load(path: str, name: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def g1_thermal_renderer() -> None:
    mod = _load(f"{REF}/thermo_nerf/thermal_nerf/thermal_renderer.py", "ref_thermal_renderer")
    g = torch.Generator().manual_seed(1234)
    R, S = 64, 48
    thermal = torch.rand(R, S, 1, generator=g) * 1.4 - 0.2  # exercises the eval clamp on both sides
    w = torch.rand(R, S, 1, generator=g)
    w = w / w.sum(dim=1, keepdim=True) * torch.rand(R, 1, 1, generator=g)  # sum(w) in (0,1)
    w[0] = 0.0  # sum(w) == 0 -> pure background
    w[1] = w[1] * 0 + 1.0 / 24.0  # sum(w) == 2 > 1
    thermal[2, 5, 0] = float("nan")  # eval: nan_to_num
    thermal[3, -1, 0] = float("inf")
    thermal[4, 7, 0] = float("-inf")
    rend = mod.ThermalRenderer()
    rend.train()
    out_train = rend(thermal.clone(), w.clone())
    rend.eval()
    out_eval = rend(thermal.clone(), w.clone())
    np.savez(os.path.join(OUT, "thermal_renderer.npz"), thermal=thermal.numpy(), weights=w.numpy(),
             out_train=out_train.numpy(), out_eval=out_eval.numpy())
    print("G1", out_train.shape, out_eval.shape)


def g2_mae_thermal() -> None:
    mod = _load(f"{REF}/thermo_nerf/thermal_nerf/thermal_metrics.py", "ref_thermal_metrics")
    bounds = json.load(open(f"{REF}/tests/data/thermal/temperature_bounds.json"))
    print("bounds", bounds)
    vals = list(bounds.values()) if isinstance(bounds, dict) else list(bounds)
    flat = []
    for v in vals:
        if isinstance(v, (int, float)):
            flat.append(float(v))
        elif isinstance(v, dict):
            flat.extend(float(x) for x in v.values() if isinstance(x, (int, float)))
    tmax, tmin = max(flat), min(flat)
    g = torch.Generator().manual_seed(4321)
    gt = torch.rand(1, 1, 32, 32, generator=g)
    pred = (gt + 0.1 * torch.randn(1, 1, 32, 32, generator=g)).clamp(0, 1)
    res = {}
    for cold in (False, True):
        for thr in (None, 0.5):
            res[f"cold{int(cold)}_thr{thr}"] = float(mod.mae_thermal(gt, pred, cold, tmax, tmin, threshold=thr))
    np.savez(os.path.join(OUT, "mae_thermal.npz"), gt=gt.numpy(), pred=pred.numpy(), tmax=tmax, tmin=tmin,
             **{k: np.float64(v) for k, v in res.items()})
    print("G2", tmax, tmin, res)


def g3_oracle_regression() -> None:
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from oracle import hotpath as H
    from thermo_nerf_amd.synthetic import counter_uniform

    out = {}
    for L, lo, hi in ((16, 16, 2048), (5, 16, 128), (5, 16, 256)):
        out[f"scalings_{L}_{lo}_{hi}"] = H.hash_scalings(L, lo, hi).numpy()
    coords = (counter_uniform(3000, 7) * 2048).to(torch.int32).view(1000, 1, 3).expand(-1, 5, -1).contiguous()
    out["hash_coords"] = coords[:, 0].numpy()
    out["hash_idx_T17_L5"] = H.hash_fn(coords, 2**17, torch.arange(5) * 2**17).numpy()
    table = (counter_uniform(5 * 2**12 * 2, 8) * 2 - 1).view(-1, 2)
    p = counter_uniform(3000, 9).view(1000, 3)
    out["enc_p"] = p.numpy()
    out["enc_out_T12_L5"] = H.hash_encode(p, table, H.hash_scalings(5, 16, 128), 12).numpy()
    x = torch.tensor([[0.5, -0.2, 0.1], [1.0, 0.0, 0.0], [2.0, -1.0, 0.5], [1e3, 2e2, -5e2], [0.0, 0.0, 0.0]])
    out["contract_in"], out["contract_out"] = x.numpy(), H.contract_inf(x).numpy()
    for near in (0.0, 0.05):
        s = H.sample_initial(torch.full((2, 1), near), torch.full((2, 1), 1000.0), 256, None)
        out[f"piecewise_ends_near{near}"] = s.ends[0, :, 0].numpy()
    w = counter_uniform(32 * 256, 11).view(32, 256, 1) ** 8
    w[0] = 0.0
    prev = H.sample_initial(torch.zeros(32, 1), torch.full((32, 1), 1000.0), 256, None)
    s = H.sample_pdf(prev, w, 96, None)
    out["pdf_w"] = w[..., 0].numpy()
    out["pdf_bins"] = torch.cat([s.spacing_starts[..., 0], s.spacing_ends[:, -1:, 0]], -1).numpy()
    np.savez_compressed(os.path.join(OUT, "oracle_regression.npz"), **out)
    print("G3", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    os.makedirs(OUT, exist_ok=True)
    g3_oracle_regression()  # before the stubs go in: uses nothing from nerfstudio
    _stub_modules()
    g1_thermal_renderer()
    g2_mae_thermal()
