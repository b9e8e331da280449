"""Diagnostic: parity of the eval render on TRAINED weights (the 1000-step config-1 problem of tests/helpers.py) — every HIP
kernel form against the fp32 CPU oracle, with the oracle's own fp64 run as the yardstick (how far two correct fp32
implementations of this field can be from each other).  usage (GPU box): python tools/trained_parity.py [steps]"""
import copy
import sys

import torch

sys.path.insert(0, ".")
from oracle import hotpath as H  # noqa: E402
from tests import helpers  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402

DEV = "cuda"


def to64(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    torch.set_num_threads(min(16, torch.get_num_threads()))  # the oracle's small ops crawl on all 256 threads of the GPU box
    prob = helpers.config1_problem()
    gm = copy.deepcopy(prob["model"]).to(DEV)
    gm.train()
    params = [p for n, p in gm.named_parameters() if not n.startswith("camera_optimizer")]
    opt = torch.optim.Adam(params, lr=1e-2, eps=1e-15, fused=True)
    o, d, cam = prob["o"].to(DEV), prob["d"].to(DEV), prob["cam"].to(DEV)
    img, th, idx = prob["image"].to(DEV), prob["thermal"].to(DEV), prob["idx"].to(DEV)
    jitter = prob["jitter"].squeeze(-1).to(DEV)
    h = prob["held_out"]

    def report(tag):
        sd = {**prob["sd"], **{k: v.detach().cpu() for k, v in gm.state_dict().items() if k in prob["sd"]}}
        # nerfstudio's sampler applies its current proposal-weight anneal in eval renders too
        anneal = float(gm.proposal_sampler._anneal)
        ref32 = H.get_outputs(sd, h["o"], h["d"], None, prob["ocfg"], anneal=anneal)
        ref64 = H.get_outputs(to64(sd), h["o"].double(), h["d"].double(), None, prob["ocfg"], anneal=anneal)
        rows = {"oracle fp32": ref32}
        was = gm.training
        gm.eval()
        for name, kw in (("hip fused lane_ray", dict(fused=True, kernel_family="lane_ray", use_mfma=True)),
                         ("hip fused ray_per_wave", dict(fused=True, kernel_family="ray_per_wave", use_mfma=True)),
                         ("hip fused valu", dict(fused=True, kernel_family="ray_per_wave", use_mfma=False)),
                         ("hip fused f16x3", dict(fused=True, kernel_family="lane_ray", use_mfma=True, mlp_precision="f16x3")),
                         ("hip modular (torch op order)", dict(fused=False))):
            saved = {k: getattr(gm.config, k) for k in kw}
            for k, v in kw.items():
                setattr(gm.config, k, v)
            gm.invalidate_prepared()
            with torch.no_grad():
                out = gm(RayBundle(origins=h["o"].to(DEV), directions=h["d"].to(DEV),
                                   camera_indices=torch.zeros((h["o"].shape[0], 1), dtype=torch.long, device=DEV)))
            rows[name] = {k: v.cpu() for k, v in out.items() if torch.is_tensor(v)}
            for k, v in saved.items():
                setattr(gm.config, k, v)
        gm.invalidate_prepared()
        gm.train(was)
        print(f"--- {tag}: mean |x - oracle fp64| (rgb, thermal, accumulation)   and vs oracle fp32")
        for name, out in rows.items():
            e64 = [float((out[k].double().reshape(-1) - ref64[k].reshape(-1)).abs().mean()) for k in ("rgb", "thermal", "accumulation")]
            e32 = [float((out[k].reshape(-1) - ref32[k].reshape(-1)).abs().mean()) for k in ("rgb", "thermal", "accumulation")]
            print(f"{name:32s} vs fp64 " + " ".join(f"{e:.2e}" for e in e64) + "   vs fp32 " + " ".join(f"{e:.2e}" for e in e32))

    report("initial weights")
    for i in range(steps):
        gm.set_step(i)
        ix = idx[i]
        rb = gm.collider(RayBundle(origins=o[ix], directions=d[ix], camera_indices=cam[ix]))
        out = TR.get_outputs_train(gm, rb, jitter=jitter[i].contiguous())
        b = {"image": img[ix], "thermal": th[ix]}
        loss = sum(gm.get_loss_dict(out, b, gm.get_metrics_dict(out, b)).values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if i + 1 in (20, 100, 300):
            report(f"after {i + 1} steps")
    report(f"after {steps} steps")


if __name__ == "__main__":
    main()
