"""In-step cost of the field's table-gradient scatter, level by level: the training step of tools/train_bench.py with
training.hash_encode_bwd replaced by one tn_hash_encode_bwd_levels launch per level (global atomics), HIP events around each.
usage: python tools/scatter_levels.py [samples=48] [steps=60]"""
import collections
import runpy
import sys

import torch

sys.path.insert(0, ".")
S = sys.argv[1] if len(sys.argv) > 1 else "48"
steps = sys.argv[2] if len(sys.argv) > 2 else "60"
sys.argv = ["train_bench.py", "--steps", steps, "--warmup", "10", "--samples", S]
from thermo_nerf_amd import _hip  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402

orig = TR.hash_encode_bwd
rec = []
stats = []


def timed(grid, space, pos, d_enc, d_table, bucketed=False, spread=True):
    if grid.num_levels != 16:
        return orig(grid, space, pos, d_enc, d_table, bucketed, spread)
    lib = _hip.load()
    n = pos.shape[0]
    for l in range(16):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _hip.check(lib.tn_hash_encode_bwd_levels(grid, space, pos.data_ptr(), d_enc.data_ptr(), n, d_table.data_ptr(), l, l + 1,
                                                 _hip.current_stream()), "levels")
        b.record()
        rec.append((l, a, b))
    if len(stats) < 3:
        stats.append((float((d_enc.abs().sum(1) > 0).float().mean()), float((d_enc.abs().sum(1) > 1e-12).float().mean()),
                      float((pos.abs().amax(1) < 1).float().mean())))


TR.hash_encode_bwd = timed
runpy.run_path("tools/train_bench.py", run_name="__main__")
torch.cuda.synchronize()
by = collections.defaultdict(list)
for l, a, b in rec[16 * 20:]:
    by[l].append(a.elapsed_time(b) * 1e3)
tot = 0.0
for l in range(16):
    v = sum(by[l]) / len(by[l])
    tot += v
    print(f"level {l:2d}: {v:7.1f} us")
print(f"sum {tot:.1f} us per step; samples with a nonzero gradient row / > 1e-12 / inside the unit box: {stats}")
