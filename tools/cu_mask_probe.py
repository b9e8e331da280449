"""Probe: can the final level's MFMA-bound backward and its table-gradient scatter share the chip side by side if each gets its
own CUs?  (chunk-pipelining them on plain streams failed because a backward block needs a whole CU and scatter
blocks take every CU that frees up.)  Two streams made with hipExtStreamCreateWithCUMask — `--scatter-cus` CUs for the atomic
scatter (its rate is set by the memory side's atomic unit: 17 G/s from 32 CUs, 21 from 64), the rest for tn_field_bwd_fused —
and five timings on the same inputs: each kernel group alone on the whole chip, each alone on its partition, both together.
usage: python tools/cu_mask_probe.py [--samples 192] [--scatter-cus 64] [--layout low|spread]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, _hip, synthetic  # noqa: E402


def masked_stream(hip, bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    err = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if err != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {err}")
    return torch.cuda.ExternalStream(s.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--scatter-cus", type=int, default=64)
    ap.add_argument("--layout", default="low", choices=["low", "spread"],
                    help="which mask bits the scatter gets: the lowest ones, or every (256 / n)-th")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    k = a.scatter_cus
    sc_bits = list(range(k)) if a.layout == "low" else list(range(0, 256, 256 // k))[:k]
    bw_bits = [b for b in range(256) if b not in set(sc_bits)]
    s_sc, s_bw = masked_stream(hip, sc_bits), masked_stream(hip, bw_bits)
    main_s = torch.cuda.current_stream()

    model = ThermalNerfModel(ThermalNerfModelConfig(num_nerf_samples_per_ray=a.samples), metadata={"thermal": []},
                             scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    model.to(dev).train()
    lib = _hip.load()
    R, S = a.rays, a.samples
    N = R * S
    g = torch.Generator(device="cpu").manual_seed(0)
    # samples along real rays (consecutive samples of a ray are neighbours, as in a step), rays from random pixels
    o, d, cam64 = synthetic.random_pixel_rays(R)
    t = torch.sort(torch.rand(R, S, generator=g) * 1.6 + 0.05, dim=1).values
    pos = (o[:, None, :] + d[:, None, :] * t[..., None]).reshape(N, 3).contiguous().to(dev)
    dirs, cam = d.to(dev), cam64.reshape(R).to(torch.int32).to(dev)
    fld = model.field.c_struct(prepare=True, dense=False)
    f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    st = _hip.current_stream()
    ray_bias = f32(R, 64)
    _hip.check(lib.tn_ray_head_fwd(fld, dirs.data_ptr(), cam.data_ptr(), R, ray_bias.data_ptr(), st), "tn_ray_head_fwd")
    enc, sel, dens, rgb, th = f32((N + 63) // 64 * 64, 32), f32(N), f32(N), f32(N, 3), f32(N, 1)
    _hip.check(lib.tn_field_fwd_train(fld, pos.data_ptr(), ray_bias.data_ptr(), R, S, enc.data_ptr(), sel.data_ptr(),
                                      dens.data_ptr(), rgb.data_ptr(), th.data_ptr(), None, None, st), "fwd")
    g_rgb, g_th, g_dens = torch.randn(N, 3, device=dev) * 1e-3, torch.randn(N, device=dev) * 1e-3, torch.randn(N, device=dev) * 1e-3
    g_enc, g_ray, g_pos = f32(N, 32), torch.zeros(R, 64, device=dev), f32(N, 3)
    grads = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    gr = _hip.tn_field_grads()
    names = {"base0": "field.mlp_base.mlp.layers.0", "base1": "field.mlp_base.mlp.layers.1", "head0": "field.mlp_head.layers.0",
             "head1": "field.mlp_head.layers.1", "head2": "field.mlp_head.layers.2", "th0": "field.mlp_thermal.layers.0",
             "th1": "field.mlp_thermal.layers.1", "thead": "field.field_head_thermal.net"}
    for kk, nme in names.items():
        setattr(gr, kk + "_w", grads[nme + ".weight"].data_ptr())
        if kk != "head0":
            setattr(gr, kk + "_b", grads[nme + ".bias"].data_ptr())
    ws = torch.empty(lib.tn_field_bwd_fused_workspace_bytes(R, S), dtype=torch.uint8, device=dev)
    d_table = grads["field.mlp_base.encoder.hash_table"]
    g_enc2 = torch.randn(N, 32, device=dev) * 1e-4  # what the scatter reads while the backward writes g_enc

    def bwd(stream):
        _hip.check(lib.tn_field_bwd_fused(fld, R, S, enc.data_ptr(), sel.data_ptr(), None, ray_bias.data_ptr(), rgb.data_ptr(),
                                          g_rgb.data_ptr(), g_th.data_ptr(), g_dens.data_ptr(), 1, -15.0, 1, g_enc.data_ptr(),
                                          g_ray.data_ptr(), pos.data_ptr(), None, g_pos.data_ptr(), C.byref(gr), ws.data_ptr(), ws.numel(),
                                          stream.cuda_stream), "bwd")

    def scatter(stream):
        _hip.check(lib.tn_hash_encode_bwd_levels(fld.grid, fld.space, pos.data_ptr(), g_enc2.data_ptr(), N, d_table.data_ptr(), 0, 16,
                                                 stream.cuda_stream), "tn_hash_encode_bwd_levels")

    def wall(jobs):
        """jobs: [(fn, stream)], all started together; ms until the last one ends"""
        best = 1e9
        for it in range(a.iters + 2):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            ends = []
            for fn, s in jobs:
                s.wait_event(e0)
                fn(s)
                e = torch.cuda.Event(enable_timing=True)
                e.record(s)
                ends.append(e)
            torch.cuda.synchronize()
            if it >= 2:
                best = min(best, max(e0.elapsed_time(e) for e in ends))
        return best * 1e3

    print(f"R {R} S {S}: scatter stream {len(sc_bits)} CUs ({a.layout}), backward stream {len(bw_bits)} CUs; us, best of {a.iters}")
    t_b, t_s = wall([(bwd, main_s)]), wall([(scatter, main_s)])
    print(f"  whole chip:   backward {t_b:8.1f}   atomic scatter (16 levels) {t_s:8.1f}   one after the other {t_b + t_s:8.1f}")
    t_bm, t_sm = wall([(bwd, s_bw)]), wall([(scatter, s_sc)])
    print(f"  partitions:   backward {t_bm:8.1f}   atomic scatter             {t_sm:8.1f}   (each alone on its CUs)")
    t_both = wall([(bwd, s_bw), (scatter, s_sc)])
    print(f"  side by side: {t_both:8.1f}   (= {t_both / (t_b + t_s):.2f} of one after the other)")
    t_plain = wall([(bwd, main_s), (scatter, torch.cuda.Stream())])
    print(f"  side by side on two UNMASKED streams: {t_plain:8.1f}")


if __name__ == "__main__":
    main()
