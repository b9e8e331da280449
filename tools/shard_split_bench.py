"""Sample-split tiles on the shards of a frame: the run one rank owns at N = 1, 2, 4, 8 (distributed.ray_block) rendered with
sample_split = 1 (whole tiles) and with the library's choice for that run size, + the reference's 65 536-ray chunk as ONE call.
usage (GPU box): python tools/shard_split_bench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from thermo_nerf_amd import distributed as D  # noqa: E402
from thermo_nerf_amd import synthetic  # noqa: E402


class A:
    dense_mb, field_dense_mb, no_mfma, precision, early_eps, weights = 64, 16, False, "f32", 0.0, "scene"


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    for H_, W_, S in ((800, 800, 192), (1080, 1920, 48), (800, 800, 64)):
        model, cfg, _, engine = bench.build_render(dev, S, bench.REF_CHUNK, A)
        o3, d3, _ = synthetic.orbit_camera_rays(H_, W_, view=3)
        o, d = o3.reshape(-1, 3).contiguous().to(dev), d3.reshape(-1, 3).contiguous().to(dev)
        n = H_ * W_
        t1 = None
        for N in (1, 2, 4, 8, 16):
            a, b = D.ray_block(n, N // 2, N)
            oa, da = o[a:b].contiguous(), d[a:b].contiguous()
            out = engine.allocate_outputs(b - a, dev)
            row = []
            for k in sorted({1, engine.shard_sample_split(b - a)} | ({2, 4, 8} if N >= 4 else set())):
                def once(k=k):
                    _, bounds = engine.render_shard(oa, da, a, n, out=out, sample_split=k)
                    engine.apply_depth_bounds(out, a, bounds)
                row.append((k, timed(once)))
            if N == 1:
                t1 = row[0][1]
            print("%dx%d S=%d N=%d shard %d rays: " % (W_, H_, S, N, b - a) +
                  "  ".join("k=%d %.3f ms (eff %.2f)" % (k, ms, t1 / (N * ms)) for k, ms in row) + "   library's choice k=%d" % engine.shard_sample_split(b - a), flush=True)
        # ONE 65 536-ray call through the plugin surface's kernels (what the reference's eval loop issues per chunk)
        oc, dc = o[:65536].contiguous(), d[:65536].contiguous()
        outc = engine.allocate_outputs(65536, dev)
        for k in (1, 2, 4, None):
            print("   65 536-ray call, sample_split=%s: %.3f ms" % (k, timed(lambda: engine.render(oc, dc, out=outc, sample_split=k))), flush=True)
        del model, engine
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
