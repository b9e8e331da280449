"""cProfile of the HOST side of the training step (tools/train_bench.py's step) on the GPU box: where the Python time between the
forward's launch and the backward's first launch goes.  usage: python tools/host_profile.py [samples=48] [steps=300]"""
import cProfile
import io
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=S)
model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
synthetic.fill_model_(model, "scene")
model.to(dev).train()
groups = model.get_param_groups()
from thermo_nerf_amd.optim import HipAdam  # noqa: E402

opt = HipAdam([{"params": groups["fields"]}, {"params": groups["proposal_networks"]}, {"params": groups["camera_opt"], "lr": 6e-4}],
              lr=1e-2, eps=1e-15, deferred=[model.field.mlp_base.encoder.hash_table])
model.config.deferred_table_update = True
o, d, cam = (t.to(dev) for t in synthetic.random_pixel_rays(4096))
g = torch.Generator().manual_seed(0)
batch = {"image": torch.rand(4096, 3, generator=g).to(dev), "thermal": torch.rand(4096, 1, generator=g).to(dev)}
torch.set_num_threads(1)


def step(i):
    model.set_step(i)
    out = model(RayBundle(origins=o, directions=d, camera_indices=cam))
    loss = TR.total_loss(model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch)))
    opt.zero_grad(set_to_none=True)
    TR.backward_total(loss)
    opt.step()


for i in range(30):
    step(5000 + i)
torch.cuda.synchronize()
# the autograd engine runs Function.backward on its own device thread: a second profiler, switched on inside RenderTrain.backward
bw = cProfile.Profile()
_orig = TR.RenderTrain.backward


def _profiled_backward(ctx, *grads):
    bw.enable()
    try:
        return _orig(ctx, *grads)
    finally:
        bw.disable()


TR.RenderTrain.backward = staticmethod(_profiled_backward)
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(5030 + i)
pr.disable()
torch.cuda.synchronize()
for prof, n in ((pr, 30), (bw, 60)):
    for key in ("cumulative", "tottime"):
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats(key).print_stats(n)
        print(buf.getvalue().replace(ROOT + "/", ""))
