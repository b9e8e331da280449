"""Scratch timing of tn_hash_encode_bwd_input on the full-size field grid.  usage: [THERMONERF_HIP_LIB=ab_x.so] python tools/posgrad_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, _hip, synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model = ThermalNerfModel(ThermalNerfModelConfig(), metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    model.to(dev).train()
    fld = model.field.c_struct(prepare=False, dense=False)
    lib = _hip.load()
    g = torch.Generator(device=dev).manual_seed(0)
    for n in (4096 * 48, 4096 * 192):
        # positions along rays (neighbouring samples share coarse cells), like a training batch
        o = (torch.rand(4096, 1, 3, device=dev, generator=g) - 0.5) * 1.2
        d = torch.nn.functional.normalize(torch.randn(4096, 1, 3, device=dev, generator=g), dim=-1)
        t = torch.sort(torch.rand(4096, n // 4096, 1, device=dev, generator=g), dim=1).values * 3.0
        pos = (o + d * t).reshape(-1, 3).contiguous()
        ge = torch.randn(n, 32, device=dev, generator=g)
        out = torch.empty(n, 3, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            _hip.check(lib.tn_hash_encode_bwd_input(fld.grid, fld.space, pos.data_ptr(), ge.data_ptr(), n, out.data_ptr(), s), "x")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            _hip.check(lib.tn_hash_encode_bwd_input(fld.grid, fld.space, pos.data_ptr(), ge.data_ptr(), n, out.data_ptr(), s), "x")
        b.record()
        torch.cuda.synchronize()
        print(f"{os.environ.get('THERMONERF_HIP_LIB', 'lib')}: n {n}: {a.elapsed_time(b) / 50 * 1e3:.1f} us  checksum {float(out.double().sum()):.6e}")


if __name__ == "__main__":
    main()
