"""Throughput of the per-sample plugin-surface entry points (Field.density_fn / get_density / get_outputs callers, e.g. export
tools querying densities on a grid).  usage: python tools/plugin_bench.py [--points 8388608]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, _hip, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1 << 23)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = ThermalNerfModel(ThermalNerfModelConfig(), metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
synthetic.fill_model_(model, "scene")
model.to(dev).eval()
lib = _hip.load()
n = a.points
pos = (torch.rand(n, 3, device=dev) * 2 - 1) * 1.5
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
dens, geo = torch.empty(n, device=dev), torch.empty(n, 15, device=dev)
rgb, th = torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
fld = model.field.c_struct()
p0 = model.proposal_networks[0].c_struct()
st = _hip.current_stream()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


for name, fn in (
    ("tn_density_fwd (proposal net 0)", lambda: _hip.check(lib.tn_density_fwd(p0, pos.data_ptr(), n, dens.data_ptr(), st), "d")),
    ("tn_field_density_fwd", lambda: _hip.check(lib.tn_field_density_fwd(fld, pos.data_ptr(), n, dens.data_ptr(), geo.data_ptr(), st), "fd")),
    ("tn_field_heads_fwd", lambda: _hip.check(lib.tn_field_heads_fwd(fld, dirs.data_ptr(), geo.data_ptr(), None, n, 0, rgb.data_ptr(), th.data_ptr(), st), "fh")),
):
    dt = timed(fn)
    print(f"{name:34s} {dt * 1e3:8.2f} ms for {n} points = {n / dt / 1e6:8.1f} M points/s")
