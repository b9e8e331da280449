"""Render time vs ray count for the two kernel families (lane = ray / one ray per wave).
usage: python tools/small_r_bench.py [auto|lane_ray|ray_per_wave]   (config.kernel_family; default auto)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd.engine import RayRenderEngine  # noqa: E402

dev = torch.device("cuda:0")
cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=64, kernel_family=sys.argv[1] if len(sys.argv) > 1 else "auto")
model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
synthetic.fill_model_(model, "scene")
model.to(dev).eval()
o, d, _ = synthetic.orbit_camera_rays(800, 800, view=1)
o, d = o.reshape(-1, 3).to(dev), d.reshape(-1, 3).to(dev)
for R in (1024, 4096, 16384, 32768, 65536, 131072, 262144, 640000):
    eng = RayRenderEngine(model, chunk=R, streams=1)
    oo, dd = o[:R].contiguous(), d[:R].contiguous()
    out = eng.allocate_outputs(R, dev)
    for _ in range(3):
        eng.render(oo, dd, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.render(oo, dd, out=out, record_events=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    p, m = eng.drain_timings()
    print(f"R {R:7d}: {dt * 1e3:8.3f} ms  proposal {sum(p) / n:7.3f}  field {sum(m) / n:7.3f}  ({R / dt / 1e6:6.2f} M rays/s)")
