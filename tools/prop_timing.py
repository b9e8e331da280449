"""Read the per-section cycle counters of the proposal `timing` build (ab_ptiming.so, see the session notes in DESIGN.md)."""
import copy, sys, torch
sys.path.insert(0, ".")
from tests import helpers
from thermo_nerf_amd import synthetic
from thermo_nerf_amd.engine import RayRenderEngine
S = 64
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
names = ["edge+position", "hash encode", "MLP", "weights+stores"]
n = c[6]
print("samples (wave level)", n, "tiles", c[7])
tot = sum(c[:4])
for nm, v in zip(names, c[:4]):
    print(f"{nm:16s} {v / n:9.0f} cycles/sample  {100 * v / tot:5.1f}%")
print(f"per sample total {tot / n:9.0f};  pdf walks per tile {c[4] / c[7]:9.0f};  tile total {c[5] / c[7]:9.0f} cycles")
