"""Diagnostic for BASELINE config 3 (bench.py `variants.train_config3_S192`): the same Trainer run with the held-out quality and
the loss terms printed along the way.  usage (GPU box): python tools/config3_fit.py [--steps 30000] [--samples 192] [--res 800]
[--views 36] [--camera-opt SO3xR3|off] [--lr 1e-2] [--avg-init-density 1.0] [--scene analytic|box]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd.cameras import frame_metrics  # noqa: E402
from thermo_nerf_amd.trainer import RayDataset, Trainer, TrainerConfig, render_view  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30000)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--views", type=int, default=36)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--camera-opt", default="SO3xR3")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--scene", default="room", choices=["room", "backdrop"])
    ap.add_argument("--marks", default="0,100,250,500,1000,2000,3500,5000,7500,10000,15000,20000,25000,30000")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    V = a.views
    views = list(range(V))
    if a.scene == "room":
        train_cams = synthetic.spiral_cameras(a.res, a.res, V)
        test_cams = synthetic.spiral_cameras(a.res, a.res, 2, phase=0.5)
    else:
        train_cams = synthetic.orbit_cameras(a.res, a.res, views, num_views=V, elevation_deg=[(-10.0, 20.0, 50.0)[v % 3] for v in views])
        test_cams = synthetic.orbit_cameras(a.res, a.res, [0.5, V / 2 + 0.5], num_views=V, elevation_deg=[5.0, 35.0])
    scene = synthetic.analytic_room_scene if a.scene == "room" else synthetic.analytic_scene

    def truth(cams, i):
        rb = cams.generate_rays(i, device=dev)
        return scene(rb.origins, rb.directions)

    imgs, ths = zip(*[truth(train_cams, i) for i in range(V)])
    ds = RayDataset.from_images(train_cams, imgs, ths, dev)
    held = [truth(test_cams, i) for i in range(2)]
    kw = {}
    if a.small:
        kw = dict(log2_hashmap_size=15, proposal_net_args_list=[
            {"hidden_dim": 16, "log2_hashmap_size": 13, "num_levels": 5, "max_res": 128, "use_linear": False},
            {"hidden_dim": 16, "log2_hashmap_size": 13, "num_levels": 5, "max_res": 256, "use_linear": False}])
    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=a.samples, eval_num_rays_per_chunk=a.res * a.res,
                                 camera_optimizer_mode=a.camera_opt, **kw)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=V)
    synthetic.fill_model_(model, "init")
    model.to(dev)
    tr = Trainer(model, ds, TrainerConfig(max_num_iterations=a.steps, train_num_rays_per_batch=a.rays))

    def evaluate():
        rows = []
        for i in range(2):
            out = render_view(model, test_cams, i, dev)
            m = frame_metrics(out, held[i][0], held[i][1], 33.085, 13.896)
            m["acc"] = float(out["accumulation"].mean())
            m["th_mean"] = float(out["thermal"].mean())
            m["depth"] = float(out["depth"].median())
            rows.append(m)
        return {k: sum(r[k] for r in rows) / 2 for k in rows[0]}

    marks = [int(x) for x in a.marks.split(",") if int(x) <= a.steps]
    t0 = time.perf_counter()
    for m in marks:
        if m > tr.step:
            n = m - tr.step
            t = time.perf_counter()
            tr.train(n)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / n * 1e3
        else:
            ms = 0.0
        # one more (untimed) iteration's loss terms, without stepping
        model.train()
        rb, batch = ds.sample(a.rays, tr.generator)
        model.set_step(tr.step)
        out = model(rb)
        md = model.get_metrics_dict(out, batch)
        ld = model.get_loss_dict(out, batch, md)
        q = evaluate()
        pose = model.camera_optimizer.pose_adjustment.detach().abs().max().item() if a.camera_opt != "off" else 0.0
        print("step %6d  %.3f ms/step | held-out psnr %.2f dB thermal mae %.3f degC (acc %.3f, thermal mean %.3f, median depth %.3f) | "
              "train psnr %.2f | %s | max |pose| %.2e" % (
                  tr.step, ms, q["psnr"], q["mae_thermal"], q["acc"], q["th_mean"], q["depth"], float(md["psnr"]),
                  " ".join("%s %.2e" % (k, float(v.detach())) for k, v in ld.items()), pose), flush=True)
    print("total %.1f s" % (time.perf_counter() - t0))


if __name__ == "__main__":
    main()
