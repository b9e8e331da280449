"""Which Python lines cause fill / zero kernels in one training step (TorchDispatchMode + traceback)."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic
from thermo_nerf_amd.rays import RayBundle

counts = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("zeros", "fill", "zero_", "full", "ones", "clone", "copy_", "mul.", "sub.", "add.", "neg", "div.")):
            fr = [f for f in traceback.extract_stack() if ("thermo_nerf_amd" in f.filename or "fill_sources" in f.filename)
                  and f.name != "__torch_dispatch__"]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(autograd/optimizer)"
            counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))

dev = torch.device("cuda:0")
cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=48)
model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
synthetic.fill_model_(model, "scene"); model.to(dev).train()
groups = model.get_param_groups()
opt = torch.optim.Adam([{"params": groups["fields"]}, {"params": groups["proposal_networks"]}, {"params": groups["camera_opt"], "lr": 6e-4}], lr=1e-2, eps=1e-15, fused=True)
g = torch.Generator().manual_seed(0)
o, d, _ = synthetic.orbit_camera_rays(64, 64, view=1)
o, d = o.reshape(-1, 3).contiguous().to(dev), d.reshape(-1, 3).contiguous().to(dev)
R = o.shape[0]
cam = torch.randint(0, 8, (R, 1), generator=g).to(dev)
batch = {"image": torch.rand(R, 3, generator=g).to(dev), "thermal": torch.rand(R, 1, generator=g).to(dev)}
def step(i):
    model.set_step(i)
    out = model(RayBundle(origins=o, directions=d, camera_indices=cam))
    loss = sum(model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch)).values())
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for i in range(3): step(5001 + i)
with Spy():
    step(5004)   # a non-update step
for (name, where), c in sorted(counts.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{c:3d}  {name:28s} {where}")
