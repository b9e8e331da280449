"""Read the per-section cycle counters of the `timing` ablation build (tools/ablate.py timing)."""
import copy, sys, torch
sys.path.insert(0, ".")
from tests import helpers
from thermo_nerf_amd import synthetic
from thermo_nerf_amd.engine import RayRenderEngine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model, _, _ = helpers.build("scene", S, small=False)
gm = copy.deepcopy(model).to("cuda:0").eval()
o, d, _ = synthetic.orbit_camera_rays(800, 800)
o, d = o.reshape(-1, 3).cuda(), d.reshape(-1, 3).cuda()
eng = RayRenderEngine(gm, chunk=640000)
eng.render(o, d); torch.cuda.synchronize()
eng._ws.zero_()
eng.render(o, d); torch.cuda.synchronize()
nb = ((640000 + 63) // 64) * 64 * (S + 1) * 4
off = (nb + 255) // 256 * 256
c = eng._ws[0][off + 8: off + 8 + 64].view(torch.int64).cpu().tolist()
names = ["hash(+pos)", "base L1", "base L2", "dens+colour", "thermal", "composite", "iters"]
it = c[6]
print("iterations", it)
tot = sum(c[:6])
for n, v in zip(names[:6], c[:6]):
    print(f"{n:14s} {v/it:10.0f} cycles/iter  {100*v/tot:5.1f}%")
print("total", tot / it)
