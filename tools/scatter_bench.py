"""Scratch timing of the two table-gradient scatters (global atomics | bucketed + LDS) on the full-size field grid, with the
real level scalings, all levels at the finest and all at the coarsest.  usage: [THERMONERF_HIP_LIB=ab_x.so] python tools/scatter_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, _hip, synthetic  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model = ThermalNerfModel(ThermalNerfModelConfig(), metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    model.to(dev).train()
    fld = model.field.c_struct(prepare=False, dense=False)
    real = [fld.grid.scalings[i] for i in range(16)]
    g = torch.Generator(device=dev).manual_seed(0)
    n = 4096 * 48
    o = (torch.rand(4096, 1, 3, device=dev, generator=g) - 0.5) * 1.2
    d = torch.nn.functional.normalize(torch.randn(4096, 1, 3, device=dev, generator=g), dim=-1)
    t = torch.sort(torch.rand(4096, 48, 1, device=dev, generator=g), dim=1).values * 3.0
    pos = (o + d * t).reshape(-1, 3).contiguous()
    if "--surface" in sys.argv:
        # a trained scene: the final level's samples sit within a few centimetres of the surface the ray hits (a sphere of
        # radius 0.3 seen from an orbit of radius 0.8), so every level's hot cells are shared by thousands of rays
        cam = torch.nn.functional.normalize(torch.randn(4096, 1, 3, device=dev, generator=g), dim=-1) * 0.8
        aim = torch.nn.functional.normalize(torch.randn(4096, 1, 3, device=dev, generator=g), dim=-1) * 0.3 * torch.rand(4096, 1, 1, device=dev, generator=g)
        dd = torch.nn.functional.normalize(aim - cam, dim=-1)
        b = (cam * dd).sum(-1, keepdim=True)
        disc = (b * b - ((cam * cam).sum(-1, keepdim=True) - 0.09)).clamp_min(0)
        th = -b - torch.sqrt(disc)
        t = th + torch.sort(torch.randn(4096, 48, 1, device=dev, generator=g) * 0.02, dim=1).values
        pos = (cam + dd * t).reshape(-1, 3).contiguous()
    ge = torch.randn(n, 32, device=dev, generator=g)
    if "--real" in sys.argv:  # the inputs of a real training step's field-level call (tools/time_scatter.py saves them)
        blob = torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "real_scatter_inputs.pt"))
        pos, ge = blob["pos"].to(dev), blob["ge"].to(dev)
        n = pos.shape[0]
    out = torch.zeros(16 << 19, 2, device=dev)
    cases = [("real scalings", real), ("all finest", [real[-1]] * 16), ("all level 6", [real[6]] * 16), ("all coarsest", [real[0]] * 16)]
    if "--levels" in sys.argv:
        cases = [("real scalings", real)] + [(f"all level {l}", [real[l]] * 16) for l in range(16)]
    for name, sc in cases:
        for i in range(16):
            fld.grid.scalings[i] = sc[i]
        for bucketed in (False, True):
            for _ in range(3):
                TR.hash_encode_bwd(fld.grid, fld.space, pos, ge, out, bucketed=bucketed)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                TR.hash_encode_bwd(fld.grid, fld.space, pos, ge, out, bucketed=bucketed)
            b.record()
            torch.cuda.synchronize()
            print(f"{os.environ.get('THERMONERF_HIP_LIB', 'lib')[-16:]}: {name:14s} {'bucketed' if bucketed else 'atomics '}: {a.elapsed_time(b) / 20 * 1e3:8.1f} us")
    for i in range(16):
        fld.grid.scalings[i] = real[i]


if __name__ == "__main__":
    main()
