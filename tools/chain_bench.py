"""Time tn_linear_chain_bwd alone on the shapes of the training step (colour head: 64 -> 64 -> 64 -> 3 with sigmoid on top).
usage: [THERMONERF_HIP_LIB=ab_x.so] python tools/chain_bench.py [--n 786432]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import _hip  # noqa: E402
from thermo_nerf_amd.training import linear_chain_bwd  # noqa: E402

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=786432)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n = a.n
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    dims = [(3, 64), (64, 64), (64, 64)]  # top first: (out, in)
    Ws = [rnd(o, i) * 0.1 for o, i in dims]
    bs = [rnd(o) for o, _ in dims]
    xs = [torch.relu(rnd(n, 64)), torch.relu(rnd(n, 64)), rnd(n, 64)]
    acts = [ACT_RELU, ACT_RELU, ACT_NONE]
    y = torch.sigmoid(rnd(n, 3))
    dy = rnd(n, 3)
    dWs = [torch.zeros_like(w) for w in Ws]
    dbs = [torch.zeros_like(b) for b in bs]
    dx = torch.zeros(n, 64, device=dev)
    lins = [_hip.tn_linear(w.data_ptr(), b.data_ptr(), w.shape[1], w.shape[0]) for w, b in zip(Ws, bs)]
    layers = [(lins[k], xs[k], 0, 64, acts[k], dWs[k], dbs[k]) for k in range(3)]

    def run():
        linear_chain_bwd(layers, y, ACT_SIGMOID, dy, 3, n, dx, 0, 64, False)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    flops = 2 * 2 * n * (3 * 64 + 64 * 64 + 64 * 64)
    if "timing" in os.environ.get("THERMONERF_HIP_LIB", ""):
        from thermo_nerf_amd.training import _CHAIN_WS
        ws = _CHAIN_WS[dev]
        cnt = ws[-64:].view(torch.int64).cpu().tolist()
        tiles = (n + 63) // 64 * (a.reps + 3)
        names = ["epilogue of the layer below / g0 stage (+barrier)", "stage x_j + barrier", "dW MFMAs", "bias sums", "dx MFMAs",
                 "barrier after the MFMAs"]
        print("wave-0 cycles per 64-row tile (3 layers):", {nm: round(c / tiles) for nm, c in zip(names, cnt)},
              "total", round(sum(cnt[:6]) / tiles))
    print(f"{os.environ.get('THERMONERF_HIP_LIB', 'lib')}: n {n}: {us:.1f} us per chain (incl. the dW reduce), "
          f"{flops / us / 1e6:.1f} TFLOP/s, {n * (64 * 4 + 3 * 2) * 4 / us / 1e3:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
