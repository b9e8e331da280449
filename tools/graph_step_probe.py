"""Probe: the whole optimisation step (forward + losses + backward + Adam) captured as HIP graphs (one per value of the proposal
sampler's update flag) and replayed, against the eager step.  usage: python tools/graph_step_probe.py [samples=48] [steps=300]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=S)
model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
synthetic.fill_model_(model, "scene")
model.to(dev).train()
groups = model.get_param_groups()
opt = torch.optim.Adam([{"params": groups["fields"]}, {"params": groups["proposal_networks"]},
                        {"params": groups["camera_opt"], "lr": 6e-4}], lr=1e-2, eps=1e-15, fused=True, capturable=True)
g = torch.Generator().manual_seed(0)
o, d, _ = synthetic.orbit_camera_rays(64, 64, view=1)
o, d = o.reshape(-1, 3).contiguous().to(dev), d.reshape(-1, 3).contiguous().to(dev)
R = o.shape[0]
cam = torch.randint(0, 8, (R, 1), generator=g).to(dev)
batch = {"image": torch.rand(R, 3, generator=g).to(dev), "thermal": torch.rand(R, 1, generator=g).to(dev)}
loss_out = torch.zeros((), device=dev)


def body():
    out = model(RayBundle(origins=o, directions=d, camera_indices=cam))
    loss = sum(model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch)).values())
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    loss_out.copy_(loss.detach())


def updated_flag():
    s = model.proposal_sampler
    return s._steps_since_update > s.update_sched(s._step) or s._step < 10


start = 5000
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(12):  # eager warm-up on the side stream: caches, workspaces, dynamic-LDS attributes, both update modes
        model.set_step(start + i)
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
step_no = start + 12
graphs = {}
pool = None
while len(graphs) < 2:
    model.set_step(step_no)
    flag = bool(updated_flag())
    if flag in graphs:
        graphs[flag].replay()
    else:
        gr = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(gr, pool=pool):
            body()
        pool = pool or gr.pool()
        graphs[flag] = gr
        print("captured graph for updated =", flag, flush=True)
    if flag:
        model.proposal_sampler._steps_since_update = 0
    step_no += 1
torch.cuda.synchronize()


def run(n, graphed):
    global step_no
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        model.set_step(step_no)
        if graphed:
            flag = bool(updated_flag())
            graphs[flag].replay()
            if flag:
                model.proposal_sampler._steps_since_update = 0
        else:
            body()
        step_no += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print("loss before", float(loss_out))
tg = run(steps, True)
print("loss after graphed", float(loss_out))
te = run(steps, False)
print("loss after eager", float(loss_out))
tg2 = run(steps, True)
print(f"S {S}: graphed {tg:.3f} ms/step, eager {te:.3f} ms/step, graphed again {tg2:.3f} ms/step")
