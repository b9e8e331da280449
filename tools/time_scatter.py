import sys, torch
sys.path.insert(0, ".")
sys.argv = ["train_bench.py", "--steps", "60", "--warmup", "10", "--samples", sys.argv[1] if len(sys.argv) > 1 else "48"]
from thermo_nerf_amd import training as TR
orig = TR.hash_encode_bwd
rec = []
def timed(grid, space, pos, d_enc, d_table, bucketed=False):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    orig(grid, space, pos, d_enc, d_table, bucketed)
    b.record()
    rec.append((pos.shape[0], grid.num_levels, a, b, pos, d_enc))
TR.hash_encode_bwd = timed
import runpy
runpy.run_path("tools/train_bench.py", run_name="__main__")
torch.cuda.synchronize()
import collections
by = collections.defaultdict(list)
for n, L, a, b, pos, ge in rec[20:]:
    by[(n, L)].append(a.elapsed_time(b) * 1e3)
for k, v in by.items():
    print(k, "calls", len(v), "avg us %.1f" % (sum(v) / len(v)))
n, L, a, b, pos, ge = [r for r in rec if r[1] == 16][-1]
print("last main call: nonzero-gradient samples %.3f, |pos|inf<1: %.3f, <2: %.3f, max %.1f" % (
    float((ge.abs().sum(1) > 0).float().mean()), float((pos.abs().amax(1) < 1).float().mean()), float((pos.abs().amax(1) < 2).float().mean()), float(pos.abs().max())))
torch.save({"pos": pos.cpu(), "ge": ge.cpu()}, "gpurun_out/real_scatter_inputs.pt") if n * 3 * 4 + n * 32 * 4 < 40e6 else None
