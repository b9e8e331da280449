"""Where a training step's torch launches come from: 8 steps under torch.profiler, device kernels grouped by the Python
line that issued them.  usage: python tools/train_ops.py [--samples 48]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=48)
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=a.samples)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    model.to(dev).train()
    groups = model.get_param_groups()
    pg = [{"params": groups["fields"]}, {"params": groups["proposal_networks"]}, {"params": groups["camera_opt"], "lr": 6e-4}]
    opt = torch.optim.Adam(pg, lr=1e-2, eps=1e-15, fused=True)
    g = torch.Generator().manual_seed(0)
    o, d, _ = synthetic.orbit_camera_rays(64, 64, view=1)
    o, d = o.reshape(-1, 3).contiguous().to(dev), d.reshape(-1, 3).contiguous().to(dev)
    R = o.shape[0]
    cam = torch.randint(0, 8, (R, 1), generator=g).to(dev)
    batch = {"image": torch.rand(R, 3, generator=g).to(dev), "thermal": torch.rand(R, 1, generator=g).to(dev)}

    def step(i):
        model.set_step(i)
        out = model(RayBundle(origins=o, directions=d, camera_indices=cam))
        metrics = model.get_metrics_dict(out, batch)
        loss = sum(model.get_loss_dict(out, batch, metrics).values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for i in range(4):
        step(5001 + i)  # 5001..: no proposal update among the profiled steps below except one
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(a.steps):
            step(5005 + i)
        torch.cuda.synchronize()
    by_line = collections.Counter()
    t_line = collections.Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        src = next((s for s in ev.stack if "thermo_nerf_amd" in s or "train_ops" in s), "(autograd engine / optimizer)")
        key = (src.strip()[-90:], ev.name)
        by_line[key] += len(ev.kernels)
        t_line[key] += sum(k.duration for k in ev.kernels)
    print(f"launches per step by source line (S={a.samples}, {a.steps} steps)")
    for (src, name), c in sorted(by_line.items(), key=lambda kv: -kv[1])[:70]:
        print(f"{c / a.steps:7.2f} {t_line[(src, name)] / a.steps:8.1f} us  {name[:34]:34s} {src}")
    print("total launches/step from torch ops:", sum(by_line.values()) / a.steps)


if __name__ == "__main__":
    main()
