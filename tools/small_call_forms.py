import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from thermo_nerf_amd import synthetic
class A: dense_mb, field_dense_mb, no_mfma, precision, early_eps, weights = 64, 16, False, "f32", 0.0, "scene"
dev = torch.device("cuda:0")
for S in (192, 48):
    model, cfg, _, engine = bench.build_render(dev, S, bench.REF_CHUNK, A)
    o3, d3, _ = synthetic.orbit_camera_rays(800, 800, view=3)
    o, d = o3.reshape(-1,3).contiguous().to(dev), d3.reshape(-1,3).contiguous().to(dev)
    for n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "40000,65536,80000,81920,160000,320000,640000".split(","))]:
        for fam in ("lane_ray", "ray_per_wave"):
            model.config.kernel_family = fam
            oa, da = o[240000:240000+n].contiguous() if n < 400000 else o[:n], d[240000:240000+n].contiguous() if n < 400000 else d[:n]
            out = engine.allocate_outputs(n, dev)
            for _ in range(2): engine.render(oa, da, out=out)
            torch.cuda.synchronize(); engine.timings = []
            for _ in range(5): engine.render(oa, da, out=out, record_events=True)
            torch.cuda.synchronize()
            p, f = engine.drain_timings()
            print("S=%d rays %7d family %-12s proposal %.3f ms  field %.3f ms" % (S, n, fam, sum(p)/len(p), sum(f)/len(f)), flush=True)
    model.config.kernel_family = "auto"
    del model, engine; torch.cuda.empty_cache()
