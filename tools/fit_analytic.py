"""Fit of a closed-form RGB + thermal scene (synthetic.analytic_scene): a freshly initialised model is trained on views
of it with the HIP training step; held-out PSNR / thermal MAE are printed.
usage: python tools/fit_analytic.py [--steps 500] [--res 64] [--views 12] [--rays 4096] [--small]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd.cameras import psnr  # noqa: E402
from thermo_nerf_amd.trainer import RayDataset, Trainer, TrainerConfig, render_view  # noqa: E402


def build(kind, small, views, dev):
    kw = {}
    if small:
        kw = dict(log2_hashmap_size=15, proposal_net_args_list=[
            {"hidden_dim": 16, "log2_hashmap_size": 13, "num_levels": 5, "max_res": 128, "use_linear": False},
            {"hidden_dim": 16, "log2_hashmap_size": 13, "num_levels": 5, "max_res": 256, "use_linear": False}])
    cfg = ThermalNerfModelConfig(camera_optimizer_mode="off", eval_num_rays_per_chunk=1 << 16, **kw)
    m = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=views)
    synthetic.fill_model_(m, kind)
    return m.to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    views = list(range(a.views))
    train_cams = synthetic.orbit_cameras(a.res, a.res, views, num_views=a.views,
                                         elevation_deg=[(-10.0, 20.0, 50.0)[v % 3] for v in views])
    test_cams = synthetic.orbit_cameras(a.res, a.res, [0.5, a.views / 2 + 0.5], num_views=a.views, elevation_deg=[5.0, 35.0])
    def truth(cams, i):
        rb = cams.generate_rays(i, device=dev)
        return synthetic.analytic_scene(rb.origins, rb.directions)

    imgs, ths = zip(*[truth(train_cams, i) for i in range(len(train_cams))])
    test = [dict(zip(("rgb", "thermal"), truth(test_cams, i))) for i in range(len(test_cams))]
    print("target stats: rgb mean %.3f std %.3f, thermal mean %.3f std %.3f" % (
        torch.stack(imgs).mean().item(), torch.stack(imgs).std().item(), torch.stack(ths).mean().item(),
        torch.stack(ths).std().item()))
    ds = RayDataset.from_images(train_cams, imgs, ths, dev)
    student = build("init", a.small, a.views, dev)
    tr = Trainer(student, ds, TrainerConfig(train_num_rays_per_batch=a.rays))

    def evaluate():
        ps, ma = [], []
        for i in range(len(test_cams)):
            o = render_view(student, test_cams, i, dev)
            ps.append(psnr(o["rgb"], test[i]["rgb"]).item())
            ma.append((o["thermal"] - test[i]["thermal"]).abs().mean().item())
        return sum(ps) / len(ps), sum(ma) / len(ma)

    print("step 0: held-out psnr %.2f dB, thermal mae %.4f" % evaluate())
    t = time.perf_counter()
    done = 0
    for chunk in (50, 50, 100, 300, 500, 1000, 3000):
        if done >= a.steps:
            break
        n = min(chunk, a.steps - done)
        tr.train(n)
        done += n
        torch.cuda.synchronize()
        p, m = evaluate()
        print("step %d: held-out psnr %.2f dB, thermal mae %.4f  (%.1f s)" % (done, p, m, time.perf_counter() - t))


if __name__ == "__main__":
    main()
