"""timing experiment: an always-update step with one piece of the proposal levels' backward skipped (wrong gradients: timing only)"""
import runpy, sys
sys.path.insert(0, ".")
what = sys.argv[1]
S = sys.argv[2]
sys.argv = ["train_bench.py", "--steps", "240", "--warmup", "30", "--samples", S, "--ray-batch", "random", "--update-every", "0"]
from thermo_nerf_amd import training as TR
orig_heb, orig_rg = TR.hash_encode_bwd, TR._ray_grads_from_enc
if what == "no_prop_scatter":
    def heb(grid, *a, **k):
        if grid.num_levels == 5:
            return
        return orig_heb(grid, *a, **k)
    TR.hash_encode_bwd = heb
elif what == "no_prop_pose":
    TR._ray_grads_from_enc = lambda *a, **k: None
elif what == "neither":
    def heb(grid, *a, **k):
        if grid.num_levels == 5:
            return
        return orig_heb(grid, *a, **k)
    TR.hash_encode_bwd = heb
    TR._ray_grads_from_enc = lambda *a, **k: None
runpy.run_path("tools/train_bench.py", run_name="__main__")
