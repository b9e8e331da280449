"""Build a proposal-kernel section-timing variant (ab_ptiming.so) and print the per-section cycles (run on the GPU box).
usage: python tools/ptiming.py build | run"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "thermo_nerf_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-undefined-internal "
         "-Wno-pass-failed -Wno-unused-variable -shared").split()

def build():
    s = open(os.path.join(CSRC, "tn_render.hip")).read().replace('#include "tn_field_eval.h"', f'#include "{CSRC}/tn_field_eval.h"')
    def rep(a, b):
        nonlocal s
        assert s.count(a) >= 1, a
        s = s.replace(a, b, 1)
    rep("    const long long stride = (long long)gridDim.x * kWaves;\n    for (long long r = (long long)blockIdx.x * kWaves + wave; r < a.R; r += stride) {\n        const float ox = a.origins[r * 3]",
        "    const long long stride = (long long)gridDim.x * kWaves;\n    unsigned long long ts[8] = {0,0,0,0,0,0,0,0};\n    for (long long r = (long long)blockIdx.x * kWaves + wave; r < a.R; r += stride) {\n        long long t0 = clock64();\n        const float ox = a.origins[r * 3]")
    rep("        const float med0 = prop_level<DOK>(", "        long long t1 = clock64(); ts[0] += t1 - t0;\n        const float med0 = prop_level<DOK>(")
    rep("        pdf_resample(wts, binsA, P0,", "        long long t2 = clock64(); ts[1] += t2 - t1;\n        pdf_resample(wts, binsA, P0,")
    rep("        const float med1 = prop_level<DOK>(", "        long long t3 = clock64(); ts[2] += t3 - t2;\n        const float med1 = prop_level<DOK>(")
    rep("        pdf_resample(wts, binsB, P1,", "        long long t4 = clock64(); ts[3] += t4 - t3;\n        pdf_resample(wts, binsB, P1,")
    rep("        for (int j = lane; j <= S; j += 64) a.ws_spacing[tn_ws_bin(r, j, S)] = binsA[j];", "        long long t5 = clock64(); ts[4] += t5 - t4;\n        for (int j = lane; j <= S; j += 64) a.ws_spacing[tn_ws_bin(r, j, S)] = binsA[j];")
    rep("        if (lane == 0) {\n            if (a.prop_depth[0]) a.prop_depth[0][r] = med0;", "        ts[5] += clock64() - t5; ts[6] += 1;\n        if (lane == 0) {\n            if (a.prop_depth[0]) a.prop_depth[0][r] = med0;")
    # flush after the ray loop of proposal_kernel: find the end marker
    rep("        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n    }\n}\n\n// ------------------------------------------------------------------------------------------------------\n// main field + composite, lane per sample (reference form)",
        "        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n    }\n    if (lane == 0) { unsigned long long *ctr = reinterpret_cast<unsigned long long *>(a.ws_spacing + (tn_ws_bin_floats(a.R, a.S) * 4 + 255) / 256 * 64) + 1 + 8;\n        for (int q = 0; q < 8; ++q) atomicAdd(ctr + q, ts[q]); }\n}\n\n// ------------------------------------------------------------------------------------------------------\n// main field + composite, lane per sample (reference form)")
    open("/tmp/ptiming.hip", "w").write(s)
    others = [os.path.join(CSRC, f) for f in ("tn_samplers.hip", "tn_fields.hip", "tn_render_mfma.hip", "tn_render_h3.hip", "tn_prepare.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *others, "/tmp/ptiming.hip", "-o", os.path.join(ROOT, "ab_ptiming.so")], check=True)
    print("built")

def run():
    import copy, torch
    sys.path.insert(0, ROOT)
    from tests import helpers
    from thermo_nerf_amd import synthetic
    from thermo_nerf_amd.engine import RayRenderEngine
    model, _, _ = helpers.build("scene", 64, small=False)
    gm = copy.deepcopy(model).to("cuda:0").eval()
    o, d, _ = synthetic.orbit_camera_rays(800, 800)
    o, d = o.reshape(-1, 3).cuda(), d.reshape(-1, 3).cuda()
    eng = RayRenderEngine(gm, chunk=640000)
    eng.render(o, d); torch.cuda.synchronize()
    eng._ws.zero_()
    eng.render(o, d); torch.cuda.synchronize()
    nb = ((640000 + 63) // 64) * 64 * 65 * 4
    off = (nb + 255) // 256 * 256
    c = eng._ws[off + 8 + 64: off + 8 + 128].view(torch.int64).cpu().tolist()
    names = ["setup+bins0", "level0 density+weights (256)", "pdf 256->96", "level1 density+weights (96)", "pdf 96->S", "store"]
    it = c[6]; tot = sum(c[:6])
    print("rays", it)
    for n, v in zip(names, c[:6]):
        print(f"{n:32s} {v/it:9.0f} cycles/ray {100*v/tot:5.1f}%")
    print("total", tot / it)

if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
