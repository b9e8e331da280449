"""Scratch timing of the fused forward (not the contract bench; see bench.py)."""
import copy
import sys
import time

import torch

sys.path.insert(0, ".")
from tests import helpers  # noqa: E402
from thermo_nerf_amd import RayBundle, synthetic  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dense = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    model, _, _ = helpers.build("scene", S, small=False, dense_grid_budget_mb=dense)
    gm = copy.deepcopy(model).to("cuda:0").eval()
    o, d, _ = synthetic.orbit_camera_rays(800, 800)
    o, d = o.reshape(-1, 3).cuda(), d.reshape(-1, 3).cuda()
    chunk = 65536
    rb = gm.collider(RayBundle(origins=o[:chunk].contiguous(), directions=d[:chunk].contiguous()))
    with torch.no_grad():
        for _ in range(2):
            gm.get_outputs(rb)
        torch.cuda.synchronize()
        t = time.time()
        n = 5
        for _ in range(n):
            gm.get_outputs(rb)
        torch.cuda.synchronize()
        dt = (time.time() - t) / n
    print(f"S={S} dense={dense}MB chunk={chunk}: {dt*1e3:.2f} ms/chunk  {chunk/dt/1e6:.3f} Mrays/s")


if __name__ == "__main__":
    main()
