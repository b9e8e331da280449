"""Build ablated copies of libthermonerf_hip.so (timing-only experiments; results are WRONG by construction).
usage: python tools/ablate.py nohash nomlp coherent ...   -> /root/repo/ab_<name>.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "thermo_nerf_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-undefined-internal "
         "-Wno-pass-failed -Wno-unused-variable -shared").split()


def variant(name: str, src: str) -> str:
    if name == "nohash":  # lane = ray kernel: features from arithmetic, no index math, no gathers
        old = "hash_encode_pipelined<L16, LG>(a.g, px, py, pz, [&](int l, float2 f) { swap32(f.x, f.y, bt0[l], bt1[l]); });"
        assert src.count(old) == 1
        return src.replace(old, "for (int l = 0; l < L16; ++l) swap32(px * (float)l, py + pz, bt0[l], bt1[l]);")
    if name == "coherent":
        return src.replace("            // ---- hash grid: 32 features",
                           "            px = __shfl(px, 0, 64); py = __shfl(py, 0, 64); pz = __shfl(pz, 0, 64);\n"
                           "            // ---- hash grid: 32 features")
    if name == "coherent_rays":  # lane = ray kernel: every lane hashes lane 0's position (one cache line per gather)
        a = src.index("__global__ void __launch_bounds__(kBlock, 2) main_mfma_rays_kernel")
        k = src[a:]
        old = "            float bt0[16], bt1[16];\n"
        assert k.count(old) == 1
        k = k.replace(old, "            px = __shfl(px, 0, 64); py = __shfl(py, 0, 64); pz = __shfl(pz, 0, 64);\n" + old)
        return src[:a] + k
    if name == "nomlp":
        i0 = src.index("            // ---- mlp_base layer 0: 32 -> 64")
        i1 = src.index("            // ---- compositing (lane = sample)")
        return src[:i0] + """            float accb = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) accb += bt0[q] * 0.01f + bt1[q] * 0.02f;
            const float dens = mul_rn(mul_rn(a.avg, expf(accb)), sel);
            float cr = accb, cg = accb * 0.5f, cb = accb * 0.25f, th = accb * 2.0f;
""" + src[i1:]
    if name == "nothing":  # neither hash nor MLP: ray setup + compositing only
        return variant("nomlp", variant("nohash", src))
    if name == "nomlp_occ":  # no MLP and no LDS blob: 8 blocks/CU instead of 2 -> is the hash phase latency-bound?
        v = variant("nomlp", src)
        v = v.replace("for (int i = threadIdx.x; i < BLOB_FLOATS / 4; i += kBlock) dst[i] = src[i];",
                      "for (int i = threadIdx.x; i < 2048 / 4; i += kBlock) dst[i] = src[i];")
        v = v.replace("lds[(a.training ? OFF_B_C1_RAW : OFF_B_C1_EVAL) + lane]", "lds[lane]")
        v = v.replace("lds[OFF_W_SH + k * 64 + lane]", "lds[64 + lane]")
        v = v.replace("lds[OFF_W_APP + k * 64 + lane]", "lds[128 + lane]")
        v = v.replace("float *scratch = lds + OFF_SCRATCH + wave * 64;", "float *scratch = lds + 1024 + wave * 64;")
        v = v.replace("const size_t smem = (size_t)LDS_FLOATS * sizeof(float);", "const size_t smem = 8192;")
        v = v.replace("const long long cap = 256LL * 2;", "const long long cap = 256LL * 8;")
        v = v.replace("__launch_bounds__(kBlock, 2)", "__launch_bounds__(kBlock, 8)")
        return v
    if name == "nomlp_8lv":  # no MLP, only the 8 finest levels hashed
        v = variant("nomlp", src)
        return v.replace("for (int l0 = 0; l0 < L16; l0 += LG) {", "for (int l0 = 8; l0 < L16; l0 += LG) {").replace(
            "float bt0[16], bt1[16];", "float bt0[16] = {}, bt1[16] = {};")
    if name == "timing":  # per-section wave-cycle sums of main_mfma_rays_kernel -> 8 uint64 counters after minmax[0..1]
        a = src.index("__global__ void __launch_bounds__(kBlock, 2) main_mfma_rays_kernel")
        b = src.index("inline bool mfma_supported")
        k = src[a:b]
        k = k.replace("    float smin = INFINITY, smax = -INFINITY;\n",
                      "    float smin = INFINITY, smax = -INFINITY;\n    unsigned long long ts[8] = {0,0,0,0,0,0,0,0};\n", 1)
        k = k.replace("            const float st = en;\n", "            long long t0 = clock64();\n            const float st = en;\n", 1)
        k = k.replace("            f32x16 h1[2][2];\n", "            long long t1 = clock64(); ts[0] += t1 - t0;\n            f32x16 h1[2][2];\n", 1)
        k = k.replace("            float g[2][8];\n", "            long long t2 = clock64(); ts[1] += t2 - t1;\n            float g[2][8];\n", 1)
        k = k.replace("            float raw, unused;\n", "            long long t3 = clock64(); ts[2] += t3 - t2;\n            float raw, unused;\n", 1)
        k = k.replace("            {   // thermal: geo", "            long long t4 = clock64(); ts[3] += t4 - t3;\n            {   // thermal: geo", 1)
        k = k.replace("            cr = nan_to_num(cr); cg = nan_to_num(cg);", "            long long t5 = clock64(); ts[4] += t5 - t4;\n            cr = nan_to_num(cr); cg = nan_to_num(cg);", 1)
        k = k.replace("            if (a.out_w && live) a.out_w[r * S + i] = wi;\n", "            if (a.out_w && live) a.out_w[r * S + i] = wi;\n            ts[5] += clock64() - t5; ts[6] += 1;\n", 1)
        k = k.replace("    if (lane == 0 && smin <= smax) {", "    if (lane == 0) { for (int q = 0; q < 8; ++q) atomicAdd(reinterpret_cast<unsigned long long *>(a.minmax + 2) + q, ts[q]); }\n    if (lane == 0 && smin <= smax) {", 1)
        assert k.count("ts[") >= 9, k.count("ts[")
        return src[:a] + k + src[b:]
    if name == "mlponly":  # rays kernel: no hash gathers, no VALU output layers: the bare MFMA chain + relu operands
        v = variant("nohash", src)
        a = v.index("__global__ void __launch_bounds__(kBlock, 2) main_mfma_rays_kernel")
        b = v.index("inline bool mfma_supported")
        k = v[a:b]
        k = k.replace("cr = fast_sigmoid(combine_halves(out_dot_fast<0>(w3, h, x2)) + w3[192]);", "cr = x2[0][0][0] + x2[1][1][3];")
        k = k.replace("cg = fast_sigmoid(combine_halves(out_dot_fast<0>(w3 + 64, h, x2)) + w3[193]);", "cg = x2[0][1][1];")
        k = k.replace("cb = fast_sigmoid(combine_halves(out_dot_fast<0>(w3 + 128, h, x2)) + w3[194]);", "cb = x2[1][0][2];")
        k = k.replace("th = combine_halves(out_dot_fast<1>(wt, h, x2)) + wt[64];", "th = x2[0][0][5] + x2[1][1][7] + x2[0][1][9] + x2[1][0][11];")
        assert k.count("x2[1][1][7]") == 1
        return v[:a] + k + v[b:]
    if name == "train_nomfma":  # chain backward without its MFMAs: loads, LDS staging, barriers, epilogues only
        assert src.count("accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw[j], 0, 0, 0);") == 1
        src = src.replace("accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw[j], 0, 0, 0);", "accw[j][s2 & 15] += av * bv;")
        assert src.count("accd = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accd, 0, 0, 0);") == 1
        return src.replace("accd = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accd, 0, 0, 0);", "accd[s2 & 15] += av * bv;")
    if name == "train_noload":  # chain backward without HBM reads of the row tiles (first tile's registers reused)
        old = "            if (next < tiles) tile_fetch(rx[j], L.x, L.ldx, IN, next * TILE, a.n, L.vec_x);\n"
        assert src.count(old) == 1
        src = src.replace(old, "")
        old = "            if (next < tiles) fetch_top(next);\n"
        assert src.count(old) == 1
        return src.replace(old, "")
    if name == "train_nobias":  # without wave 0's serial column sums
        old = "                for (int r = 0; r < TILE; ++r) sb += gs[r * LDP + threadIdx.x];\n"
        assert src.count(old) == 1
        return src.replace(old, "                sb += gs[threadIdx.x];\n")
    if name == "train_timing":  # per-phase wave-cycle sums of linear_chain_bwd_kernel (wave 0) -> 8 uint64 after the slabs
        a = src.index("template <int NL>\n__global__ void __launch_bounds__(kBlock, 2) linear_chain_bwd_kernel")
        b = src.index("// dW[o][i] += sum_b partials[b][i][o]")
        k = src[a:b]
        def rep(old, new):
            nonlocal k
            assert k.count(old) == 1, old
            k = k.replace(old, new)
        rep("    const long long tiles = (a.n + TILE - 1) / TILE;\n",
            "    const long long tiles = (a.n + TILE - 1) / TILE;\n    unsigned long long ts[8] = {0,0,0,0,0,0,0,0}; long long tq;\n")
        rep("        __syncthreads();  // the previous tile's readers of gs / xs are done\n",
            "        tq = clock64();\n        __syncthreads();  // the previous tile's readers of gs / xs are done\n")
        rep("            tile_to_lds(rx[j], xs);\n", "            { long long t = clock64(); ts[0] += t - tq; tq = t; }\n            tile_to_lds(rx[j], xs);\n")
        rep("            const int ot = wave % n_ot, it2 = wave / n_ot;\n            if (wave < n_ot * n_it) {  // dW_j tile (ot, it2)\n",
            "            { long long t = clock64(); ts[1] += t - tq; tq = t; }\n            const int ot = wave % n_ot, it2 = wave / n_ot;\n            if (wave < n_ot * n_it) {  // dW_j tile (ot, it2)\n")
        rep("            {   // bias_j: column sums", "            { long long t = clock64(); ts[2] += t - tq; tq = t; }\n            {   // bias_j: column sums")
        rep("            const bool last = j == NL - 1;\n", "            { long long t = clock64(); ts[3] += t - tq; tq = t; }\n            const bool last = j == NL - 1;\n")
        rep("            __syncthreads();  // every MFMA operand read of gs / xs is done\n",
            "            { long long t = clock64(); ts[4] += t - tq; tq = t; }\n            __syncthreads();  // every MFMA operand read of gs / xs is done\n            { long long t = clock64(); ts[5] += t - tq; tq = t; }\n")
        rep("    // one slab [i (64 rows) | bias row][o (64)] per layer and block\n",
            "    if (threadIdx.x == 0) { for (int q = 0; q < 8; ++q) atomicAdd(reinterpret_cast<unsigned long long *>(a.partials + (size_t)kChainMax * kChainBlocks * 65 * 64) + q, ts[q]); }\n    // one slab [i (64 rows) | bias row][o (64)] per layer and block\n")
        # the epilogue (after the barrier) up to the loop end is charged to ts[0] of the next layer / tile
        src = src[:a] + k + src[b:]
        old = "size_t tn_linear_chain_bwd_workspace_bytes(void) { return (size_t)kChainMax * kChainBlocks * 65 * 64 * sizeof(float); }"
        assert src.count(old) == 1
        return src.replace(old, "size_t tn_linear_chain_bwd_workspace_bytes(void) { return (size_t)kChainMax * kChainBlocks * 65 * 64 * sizeof(float) + 64; }")
    if name == "base" or name == "train_base":
        return src
    raise SystemExit(f"unknown variant {name}")


def header_variant(name: str, dev: str) -> str:
    """variants that patch tn_device.h (the hash-grid gather of the FAST flavour)"""
    if name == "nogather":  # index arithmetic + interpolation kept, the 8 loads replaced by bit-casts of the offsets
        for k, expr in enumerate(["(x1 ^ y1 ^ z1)", "(x1 ^ y0 ^ z1)", "(x0 ^ y0 ^ z1)", "(x0 ^ y1 ^ z1)", "(x1 ^ y1 ^ z0)",
                                  "(x1 ^ y0 ^ z0)", "(x0 ^ y0 ^ z0)", "(x0 ^ y1 ^ z0)"]):
            old = f"f{k} = *reinterpret_cast<const float2 *>(tb + ({expr} & m8));"
            assert dev.count(old) == 1, old
            dev = dev.replace(old, f"f{k} = make_float2(__uint_as_float(({expr} & m8) | 0x30000000u), ox);")
        return dev
    if name == "onecorner":  # all 8 corners read the SAME entry (1/8 of the distinct addresses, same instruction count)
        for k, expr in enumerate(["(x1 ^ y1 ^ z1)", "(x1 ^ y0 ^ z1)", "(x0 ^ y0 ^ z1)", "(x0 ^ y1 ^ z1)", "(x1 ^ y1 ^ z0)",
                                  "(x1 ^ y0 ^ z0)", "(x0 ^ y1 ^ z0)"]):
            dev = dev.replace(f"tb + ({expr} & m8)", "tb + ((x0 ^ y0 ^ z0) & m8)")
        return dev
    if name in ("hash_idx", "hash_ceil"):  # index-based addressing (hash_idx), and additionally the compare-based ceil corner
        a = dev.index("        if (FAST) {\n            // byte offsets straight from the hash")
        b = dev.index("        } else {\n            const float2 *t = g.table + (size_t)l * g.tsize;")
        c = dev.index("            f7 = t[(fx ^ hcy ^ hfz) & m];\n        }\n") + len("            f7 = t[(fx ^ hcy ^ hfz) & m];\n        }\n")
        body = dev[b + len("        } else {\n"):c - len("        }\n")]
        dev = dev[:a] + "        {\n" + body + "        }\n" + dev[c:]
        if name == "hash_ceil":
            dev = dev.replace("(FAST || ox > 0.0f)", "(ox > 0.0f)").replace("(FAST || oy > 0.0f)", "(oy > 0.0f)").replace(
                "(FAST || oz > 0.0f)", "(oz > 0.0f)")
        return dev
    return dev


HEADER_VARIANTS = ("nogather", "onecorner", "hash_idx", "hash_ceil")


def main():
    """names: a tn_render_mfma.hip source variant, a header variant (applied to the field kernels), or `prop_<header variant>`
    (the same header patch applied to tn_render.hip = the proposal kernels only)."""
    for name in sys.argv[1:]:
        target = "tn_render.hip" if name.startswith("prop_") else "tn_train.hip" if name.startswith("train_") else "tn_render_mfma.hip"
        hname = name[5:] if name.startswith("prop_") else name
        inc = CSRC
        if hname in HEADER_VARIANTS:
            inc = f"/tmp/abl_inc_{name}"
            os.makedirs(inc, exist_ok=True)
            open(os.path.join(inc, "tn_device.h"), "w").write(
                header_variant(hname, open(os.path.join(CSRC, "tn_device.h")).read()).replace(
                    '#include "../../include/thermonerf_hip.h"', f'#include "{ROOT}/include/thermonerf_hip.h"'))
            open(os.path.join(inc, "tn_field_eval.h"), "w").write(open(os.path.join(CSRC, "tn_field_eval.h")).read())
        src = open(os.path.join(CSRC, target)).read().replace('#include "tn_field_eval.h"', f'#include "{inc}/tn_field_eval.h"')
        tmp = f"/tmp/abl_{name}.hip"
        open(tmp, "w").write(src if hname in HEADER_VARIANTS else variant(name, src))
        # the untouched translation units come from the regular build's objects (make in csrc/ first)
        others = [os.path.join(CSRC, "build", f.replace(".hip", ".o")) for f in
                  ("tn_samplers.hip", "tn_fields.hip", "tn_render.hip", "tn_render_mfma.hip", "tn_render_h3.hip", "tn_train.hip",
                   "tn_prepare.hip", "tn_metrics.hip") if f != target and hname not in HEADER_VARIANTS]
        if hname in HEADER_VARIANTS:
            others = [os.path.join(CSRC, f) for f in ("tn_samplers.hip", "tn_fields.hip", "tn_render.hip", "tn_render_mfma.hip",
                                                      "tn_render_h3.hip", "tn_train.hip", "tn_prepare.hip", "tn_metrics.hip") if f != target]
        out = os.path.join(ROOT, f"ab_{name}.so")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *others, tmp, "-o", out], check=True)
        print("built", out)


if __name__ == "__main__":
    main()
