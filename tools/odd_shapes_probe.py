"""Eval get_outputs on odd shapes (ray counts around the wave / tile sizes, proposal and field sample counts that are no multiple of
anything) against the oracle: max abs differences per output.   usage (GPU box): python tools/odd_shapes_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hotpath as H  # noqa: E402
from tests import helpers  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402

DEV = "cuda:0"
cases = [(1, (7, 5), 3), (2, (64, 32), 1), (63, (33, 17), 48), (65, (256, 96), 13), (127, (300, 130), 200), (257, (1, 1), 2),
         (64, (2, 3), 64), (1000, (96, 256), 7), (5, (512, 256), 256)]
for R, P, S in cases:
    try:
        cm, sd, ocfg = helpers.build("scene", S, num_proposal_samples_per_ray=P)
        gm = cm.to(DEV).eval()
        o, d = helpers.rays(40, 40, view=2)
        o, d = o[:R].contiguous(), d[:R].contiguous()
        want = H.get_outputs(sd, o, d, None, ocfg)
        with torch.no_grad():
            got = gm(RayBundle(origins=o.to(DEV), directions=d.to(DEV)))
        diffs = {k: float((got[k].cpu() - want[k]).abs().max()) for k in ("rgb", "thermal", "accumulation", "expected_depth", "depth")}
        print(R, P, S, " ".join(f"{k} {v:.2e}" for k, v in diffs.items()), flush=True)
    except Exception as e:  # noqa: BLE001
        print(R, P, S, "FAILED:", type(e).__name__, str(e)[:200], flush=True)
