"""Does the SHAPE of a 64-ray tile matter to the lane = ray kernels?  The kernels take a flat ray array and give 64 consecutive
rays to a wave — on a row-major image that is a 64 x 1 strip of pixels.  This probe feeds the same 800 x 800 frame in a permuted
order so that 64 consecutive rays form a w x h pixel block (64x1 = the frame as it is, 32x2, 16x4, 8x8), renders with the
unchanged kernels and prints the proposal / field kernel times (HIP events inside the engine) and checks that the un-permuted
frame is bitwise the frame of the plain order (rays are independent; only the call-global expected-depth clip is shared).
usage: python tools/tile_shape_probe.py [--samples 192] [--steps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from thermo_nerf_amd import synthetic  # noqa: E402


def block_order(H: int, W: int, bw: int, bh: int) -> torch.Tensor:
    """flat ray indices of an H x W row-major image, reordered so that every run of bw * bh indices is one bw x bh pixel block"""
    assert H % bh == 0 and W % bw == 0
    idx = torch.arange(H * W).reshape(H // bh, bh, W // bw, bw)
    return idx.permute(0, 2, 1, 3).reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--size", type=int, default=800)
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda:0")
    H = W = a.size
    args.chunk = H * W
    model, cfg, _, engine = bench.build_render(dev, a.samples, args.chunk, args)
    o_cpu, d_cpu, _ = synthetic.orbit_camera_rays(H, W, view=0)
    o_cpu, d_cpu = o_cpu.reshape(-1, 3), d_cpu.reshape(-1, 3)
    ref = None
    for bw, bh in ((64, 1), (32, 2), (16, 4), (8, 8), (4, 16)):
        if W % bw or H % bh:
            continue
        perm = block_order(H, W, bw, bh).to(dev)
        o, d = o_cpu.to(dev)[perm].contiguous(), d_cpu.to(dev)[perm].contiguous()
        out = engine.allocate_outputs(H * W, dev)
        el, p, m = bench.timed_frames(engine, o, d, out, a.steps, 2)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(H * W, device=dev)
        frame = {k: v[inv] for k, v in out.items()}
        same = ""
        if ref is None:
            ref = frame
        else:
            same = " | vs 64x1: " + ", ".join(
                "%s %s" % (k, "bitwise" if torch.equal(frame[k], ref[k]) else "max diff %.3g" % (frame[k] - ref[k]).abs().max().item())
                for k in ("rgb", "thermal", "depth", "accumulation"))
        print("tile %2dx%-2d S=%d: frame %.3f ms, proposal %.3f ms, field %.3f ms%s" % (
            bw, bh, a.samples, el / a.steps * 1e3, sum(p) / len(p), sum(m) / len(m), same), flush=True)


if __name__ == "__main__":
    main()
