"""Looks for single-step anomalies in the HIP path's config-1 run (tests/helpers.py::config1_problem): N runs of the first K steps,
every step's loss against the per-step median over the runs (the trajectories agree to a few percent over the first 100 steps).
usage (GPU box): python tools/config1_glitch_probe.py [runs=200] [steps=120] [config key=value ...]"""
import copy
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests import helpers  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 120
over = dict(kv.split("=") for kv in sys.argv[3:])
DEV = "cuda:0"
prob = helpers.config1_problem()
o, d, cam = prob["o"].to(DEV), prob["d"].to(DEV), prob["cam"].to(DEV)
img, th, idx = prob["image"].to(DEV), prob["thermal"].to(DEV), prob["idx"].to(DEV)
jitter = prob["jitter"].squeeze(-1).to(DEV)
all_losses, all_terms = [], []
for r in range(runs):
    gm = copy.deepcopy(prob["model"]).to(DEV)
    for k, v in over.items():
        setattr(gm.config, k, v == "True")
    gm.train()
    params = [p for n, p in gm.named_parameters() if not n.startswith("camera_optimizer")]
    opt = torch.optim.Adam(params, lr=1e-2, eps=1e-15, fused=True)
    got, terms = [], []
    for i in range(K):
        gm.set_step(i)
        ix = idx[i]
        rb = gm.collider(RayBundle(origins=o[ix], directions=d[ix], camera_indices=cam[ix]))
        out = TR.get_outputs_train(gm, rb, jitter=jitter[i].contiguous())
        b = {"image": img[ix], "thermal": th[ix]}
        ld = gm.get_loss_dict(out, b, gm.get_metrics_dict(out, b))
        loss = sum(ld.values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        got.append(loss.detach())
        terms.append(torch.stack([v.detach().reshape(()) for v in ld.values()]))
    all_losses.append(torch.stack(got).cpu().numpy())
    all_terms.append(torch.stack(terms).cpu().numpy())
    names = list(ld.keys())
L = np.stack(all_losses)
T = np.stack(all_terms)
med = np.median(L, axis=0)
dev = np.abs(L - med) / med
print("runs", runs, "steps", K, "terms", names)
print("per-step max relative deviation from the median (first 40):", " ".join(f"{x:.2f}" for x in dev.max(axis=0)[:40]))
W = int(dict(a.split("=") for a in []).get("w", 100))
bad = np.argwhere(dev[:, :100] > 0.5)
print("anomalies (run, step) with |loss - median| > 0.5 median:", len(bad))
for r, s in bad[:40]:
    print(f"  run {r} step {s}: loss {L[r, s]:.5f} median {med[s]:.5f}  terms {np.round(T[r, s], 5)} median terms {np.round(np.median(T[:, s], axis=0), 5)}"
          f"  previous step {L[r, s - 1]:.5f}/{med[s - 1]:.5f} next {L[r, min(s + 1, K - 1)]:.5f}/{med[min(s + 1, K - 1)]:.5f}")
