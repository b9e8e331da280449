"""Time the tape-free pair of the final level on its own: tn_field_fwd_train and tn_field_bwd_fused (split / one launch) on random
positions and output gradients.  usage: python tools/fused_bwd_bench.py [--rays 4096] [--samples 192] [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, _hip, synthetic  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = ThermalNerfModel(ThermalNerfModelConfig(num_nerf_samples_per_ray=a.samples), metadata={"thermal": []},
                             scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    model.to(dev).train()
    lib = _hip.load()
    R, S = a.rays, a.samples
    N = R * S
    g = torch.Generator(device="cpu").manual_seed(0)
    pos = (torch.rand(N, 3, generator=g) * 2 - 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    cam = torch.randint(0, 8, (R,), generator=g).to(torch.int32).to(dev)
    fld = model.field.c_struct(prepare=True, dense=False)
    f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    st = _hip.current_stream()
    ray_bias = f32(R, 64)
    _hip.check(lib.tn_ray_head_fwd(fld, d.data_ptr(), cam.data_ptr(), R, ray_bias.data_ptr(), st), "tn_ray_head_fwd")
    enc, sel, dens, rgb, th = f32((N + 63) // 64 * 64, 32), f32(N), f32(N), f32(N, 3), f32(N, 1)
    bo = f32(N, 16)

    jac = f32((N + 63) // 64 * 64, 96)

    def fwd(keep_base=True, keep_jac=False):
        _hip.check(lib.tn_field_fwd_train(fld, pos.data_ptr(), ray_bias.data_ptr(), R, S, enc.data_ptr(), sel.data_ptr(),
                                          dens.data_ptr(), rgb.data_ptr(), th.data_ptr(), bo.data_ptr() if keep_base else None,
                                          jac.data_ptr() if keep_jac else None, st), "fwd")

    g_rgb, g_th, g_dens = torch.randn(N, 3, device=dev) * 1e-3, torch.randn(N, device=dev) * 1e-3, torch.randn(N, device=dev) * 1e-3
    g_enc, g_ray, g_pos = f32(N, 32), torch.zeros(R, 64, device=dev), f32(N, 3)
    grads = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    gr = _hip.tn_field_grads()
    names = {"base0": "field.mlp_base.mlp.layers.0", "base1": "field.mlp_base.mlp.layers.1", "head0": "field.mlp_head.layers.0",
             "head1": "field.mlp_head.layers.1", "head2": "field.mlp_head.layers.2", "th0": "field.mlp_thermal.layers.0",
             "th1": "field.mlp_thermal.layers.1", "thead": "field.field_head_thermal.net"}
    for k, nme in names.items():
        setattr(gr, k + "_w", grads[nme + ".weight"].data_ptr())
        if k != "head0":
            setattr(gr, k + "_b", grads[nme + ".bias"].data_ptr())
    ws = torch.empty(lib.tn_field_bwd_fused_workspace_bytes(R, S), dtype=torch.uint8, device=dev)

    def bwd(split, stored=False, from_jac=False):
        _hip.check(lib.tn_field_bwd_fused(fld, R, S, enc.data_ptr(), sel.data_ptr(), bo.data_ptr() if stored else None,
                                          ray_bias.data_ptr(), rgb.data_ptr(),
                                          g_rgb.data_ptr(), g_th.data_ptr(), g_dens.data_ptr(), 1, -15.0, split, g_enc.data_ptr(),
                                          g_ray.data_ptr(), pos.data_ptr(), jac.data_ptr() if from_jac else None, g_pos.data_ptr(), C.byref(gr), ws.data_ptr(), ws.numel(), st), "bwd")

    def time(fn, *args):
        for _ in range(3):
            fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3

    flops = 33024.0 * N
    for keep, kj in ((False, False), (True, False), (True, True)):
        t = time(fwd, keep, kj)
        print(f"N {N}: tn_field_fwd_train base_out={keep} jacobian={kj} {t:8.1f} us  ({flops / t / 1e6:.1f} TF of the 1x forward)")
    for split, stored, fj in ((1, False, False), (1, True, False), (2, True, False), (2, True, True), (0, False, False)):
        t = time(bwd, split, stored, fj)
        print(f"N {N}: tn_field_bwd_fused split={split} stored_base={stored} pose_from_jacobian={fj} {t:8.1f} us  ({3 * flops / t / 1e6:.1f} TF of recompute + dx + dW = 3x forward)")


if __name__ == "__main__":
    main()
