"""What would the table-gradient scatter's run merging gain if the samples were visited in SPATIAL order?  For one training batch
(the positions of the final level, as tools/line_census.py builds them) count, per level, the runs of consecutive samples in one
grid cell — each run is what the atomic kernel's segmented wave scan turns into one group of line-atomics — in the kernels' order
(ray-major: neighbours along a ray) and after sorting the samples by the Morton code of their cell on a 2^bits grid.
Runs are cut at every 64-sample wave boundary, as in the kernel.
usage (GPU box): python tools/sorted_scatter_census.py [--samples 192] [--rays 4096] [--bits 8] [--weights scene|trained-steps N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402


def part1by2(x):
    x = x & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x


def runs_per_wave(cell):
    """runs of equal consecutive values, cut at multiples of 64"""
    n = cell.shape[0]
    brk = torch.ones(n, dtype=torch.bool, device=cell.device)
    brk[1:] = cell[1:] != cell[:-1]
    brk[::64] = True
    return int(brk.sum().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--proposal", action="store_true", help="the 256 + 96 samples of the proposal levels on their 5-level grids instead")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=a.samples)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    model.to(dev).train()
    model.set_step(5000)
    o, d, cam = (t.to(dev) for t in synthetic.random_pixel_rays(a.rays))
    with torch.no_grad():
        torch.manual_seed(0)
        out = model(RayBundle(origins=o, directions=d, camera_indices=cam))
    todo = [(-1, model.field.mlp_base.encoder)] if not a.proposal else [(0, model.proposal_networks[0].mlp_base.encoder),
                                                                        (1, model.proposal_networks[1].mlp_base.encoder)]
    for which, enc in todo:
        eucl = out["ray_samples_list"][which].eucl_bins
        mid = 0.5 * (eucl[:, :-1] + eucl[:, 1:])
        pos = (o[:, None, :] + d[:, None, :] * mid[..., None]).reshape(-1, 3)
        mag = pos.abs().amax(dim=-1, keepdim=True)
        con = torch.where(mag < 1, pos, (2 - 1 / mag) * pos / mag)
        p = ((con + 2) / 4).clamp(0, 1)
        N = p.shape[0]
        q = (p * (1 << a.bits)).long().clamp(0, (1 << a.bits) - 1)
        key = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
        order = torch.argsort(key, stable=True)
        ps = p[order]
        print(f"level set {which}: N = {N}, Morton key on a {1 << a.bits}^3 grid")
        print("level scaling   runs(ray order)  runs(sorted)  ratio   runs/sample sorted")
        tot0 = tot1 = 0
        for l, s in enumerate(enc.scalings.reshape(-1).tolist()):
            def cells(pp):
                f = torch.floor(pp * s).long()
                return (f[:, 0] * 4099 + f[:, 1]) * 4099 + f[:, 2]
            r0, r1 = runs_per_wave(cells(p)), runs_per_wave(cells(ps))
            tot0, tot1 = tot0 + r0, tot1 + r1
            print(f"{l:5d} {s:7.1f} {r0:16d} {r1:13d} {r0 / r1:6.2f} {r1 / N:8.3f}")
        print(f"all levels: {tot0} -> {tot1} ({tot0 / tot1:.2f}x)")


if __name__ == "__main__":
    main()
