import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from thermo_nerf_amd import distributed as D, synthetic
class A: dense_mb, field_dense_mb, no_mfma, precision, early_eps, weights = 64, 16, False, "f32", 0.0, "scene"
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e3
for H_, W_, S, Ns in ((1080,1920,48,(4,8)), (800,800,192,(2,8))):
    model, cfg, _, engine = bench.build_render(dev, S, bench.REF_CHUNK, A)
    o3, d3, _ = synthetic.orbit_camera_rays(H_, W_, view=3)
    o, d = o3.reshape(-1,3).contiguous().to(dev), d3.reshape(-1,3).contiguous().to(dev)
    n = H_*W_
    for N in Ns:
        for rank in (0, N//2):
            a,b = D.ray_block(n, rank, N)
            oa, da = o[a:b].contiguous(), d[a:b].contiguous()
            out = engine.allocate_outputs(b-a, dev)
            res = {}
            for rep in range(3):
                for k in (1,2,4,6,8):
                    def once(k=k):
                        _, bounds = engine.render_shard(oa, da, a, n, out=out, sample_split=k)
                        engine.apply_depth_bounds(out, a, bounds)
                    res.setdefault(k, []).append(timed(once))
            print(H_, S, "N", N, "rank", rank, " ".join("k=%d %s" % (k, "/".join("%.3f" % x for x in v)) for k,v in res.items()), flush=True)
    del model, engine; torch.cuda.empty_cache()
