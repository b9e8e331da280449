"""Cost of the atomic half of the field's table-gradient scatter by level range, on the (positions, d_enc) of a real training step:
the step of tools/train_bench.py runs with training.hash_encode_bwd hooked to keep one step's tensors; afterwards the ranges are
timed alone (HIP events, 20 repeats): tn_hash_encode_bwd_spread (levels from 0: the coarse ones through private dense copies) and
tn_hash_encode_bwd_levels (global atomics).   usage: python tools/scatter_ranges.py [samples=192]"""
import runpy
import sys

import torch

sys.path.insert(0, ".")
S = sys.argv[1] if len(sys.argv) > 1 else "192"
sys.argv = ["train_bench.py", "--steps", "12", "--warmup", "4", "--samples", S, "--ray-batch", "random"]
from thermo_nerf_amd import _hip  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402

orig = TR.hash_encode_bwd
kept = {}


def hook(grid, space, pos, d_enc, d_table, *a, **k):
    if grid.num_levels == 16:
        kept.update(grid=grid, space=space, pos=pos.clone(), d_enc=d_enc.clone(), shape=d_table.shape)
    return orig(grid, space, pos, d_enc, d_table, *a, **k)


TR.hash_encode_bwd = hook
runpy.run_path("tools/train_bench.py", run_name="__main__")
lib = _hip.load()
g, sp, pos, de = kept["grid"], kept["space"], kept["pos"], kept["d_enc"]
n = pos.shape[0]
tab = torch.zeros(kept["shape"], device=pos.device)
need = lib.tn_hash_encode_bwd_spread_workspace_bytes(g)
ws = torch.empty(need, dtype=torch.uint8, device=pos.device)
st = _hip.current_stream()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def spread(hi):
    return lambda: _hip.check(lib.tn_hash_encode_bwd_spread(g, sp, pos.data_ptr(), de.data_ptr(), n, tab.data_ptr(), 0, hi, ws.data_ptr(), need, st), "s")


def plain(lo, hi):
    return lambda: _hip.check(lib.tn_hash_encode_bwd_levels(g, sp, pos.data_ptr(), de.data_ptr(), n, tab.data_ptr(), lo, hi, st), "l")


print(f"n {n}: spread 0-8 {t(spread(8)):.0f} us | spread 0-4 {t(spread(4)):.0f} | plain 4-8 {t(plain(4, 8)):.0f} | plain 0-8 {t(plain(0, 8)):.0f} | plain 0-16 {t(plain(0, 16)):.0f}")
print("spread from 0 to L: " + " ".join(f"{L}:{t(spread(L)):.0f}" for L in range(1, 9)))
print("plain single level: " + " ".join(f"{l}:{t(plain(l, l + 1)):.0f}" for l in range(16)))
