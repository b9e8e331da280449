"""Static per-phase instruction counts of one pass (64 rays x 1 sample) of main_mfma_rays_kernel (both forms): section markers
(volatile asm comments at the anchors of the `timing` ablation) are compiled into the kernel, the ISA of the sample loop is
split at the markers and VALU / MFMA / LDS / vector-memory / scalar instructions are counted per section.  The scheduler may
move a few instructions across a marker; the totals match SQ_INSTS_* / pass of the PMC profiles within a few per cent.
usage: python tools/valu_count.py > profiles/roundN_valu_by_phase.txt"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "thermo_nerf_amd", "csrc")
MARKS = [  # (anchor in the source, section that STARTS there)
    ("            const float st = en;\n", "ray march: bin edges -> position, contraction"),
    ("            float bt0[16], bt1[16];\n", "hash grid: index math, 128 gathers, trilinear blend, re-layout"),
    ("            f32x16 h1[2][2];\n", "mlp_base layer 0 (32->64) + ReLU"),
    ("            float g[2][8];\n", "mlp_base layer 1 (64->16), geo re-layout"),
    ("            float raw, unused;\n", "density (exp), colour MLP 64->64->64->3 + sigmoids"),
    ("            {   // thermal: geo", "thermal MLP 16->64->64->1"),
    ("            cr = nan_to_num(cr); cg = nan_to_num(cg);", "compositing: weights, rgb/thermal/depth running values"),
]


def main():
    src = open(os.path.join(CSRC, "tn_render_mfma.hip")).read()
    a = src.index("__global__ void __launch_bounds__(kBlock, 2) main_mfma_rays_kernel")
    b = src.index("inline bool mfma_supported")
    k = src[a:b]
    for i, (anchor, _) in enumerate(MARKS):
        assert k.count(anchor) == 1, anchor
        k = k.replace(anchor, f'            asm volatile("; TN_SECTION {i}" ::: "memory");\n' + anchor)
    # end of the sample loop body
    end_anchor = "            if (a.out_w && live) a.out_w[r * S + i] = wi;\n"
    assert k.count(end_anchor) == 1
    k = k.replace(end_anchor, end_anchor + '            asm volatile("; TN_SECTION 99" ::: "memory");\n')
    os.makedirs("/tmp/valu_count_dir", exist_ok=True)
    tmp = "/tmp/valu_count_dir/valu_count.hip"
    open(tmp, "w").write(src[:a] + k + src[b:])
    out = "/tmp/valu_count.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-function",
                    "-Wno-undefined-internal", "-Wno-pass-failed", "-S", "--cuda-device-only", f"-I{CSRC}", tmp, "-o", out], check=True)
    text = open(out).read()
    stamp = os.path.join(ROOT, "tools", ".head_stamp")
    if os.path.exists(stamp):
        print("# source at commit", open(stamp).read().strip())
    for tag, sym, what in (("ILb1", "<true>", "the default: first 6 levels from the dense re-layout"), ("ILb0", "<false>", "all 16 levels hashed")):
        m = re.search(r"^_ZN12_GLOBAL__N_121main_mfma_rays_kernel" + tag + r"EEEvNS_8MfmaArgsE:.*?s_endpgm", text, re.S | re.M)
        assert m, "kernel not found"
        report(m.group(0).splitlines(), f"main_mfma_rays_kernel{sym} ({what})")


def report(body, title):
    counts = collections.OrderedDict()
    cur = None
    for line in body:
        t = line.strip()
        mm = re.match(r"; TN_SECTION (\d+)", t)
        if mm:
            sec = int(mm.group(1))
            cur = None if sec == 99 else sec
            if cur is not None:
                counts.setdefault(cur, collections.Counter())
            continue
        if cur is None or not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = counts[cur]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if op.startswith(("v_permlane", "v_mov_b32_dpp", "v_readlane", "v_readfirstlane")):
                c["valu_xlane"] += 1
            if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_mul_lo_u32", "v_mul_hi")):
                c["valu_quarter_rate"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    print(f"# static instruction counts of ONE pass (64 rays x 1 sample) of {title}, by phase (tools/valu_count.py)")
    print("phase,valu,of_which_cross_lane,of_which_quarter_rate,mfma,lds,vmem,salu")
    tot = collections.Counter()
    for i, (_, name) in enumerate(MARKS):
        c = counts.get(i, collections.Counter())
        tot.update(c)
        print(f"\"{name}\",{c['valu']},{c['valu_xlane']},{c['valu_quarter_rate']},{c['mfma']},{c['lds']},{c['vmem']},{c['salu']}")
    print(f"\"total\",{tot['valu']},{tot['valu_xlane']},{tot['valu_quarter_rate']},{tot['mfma']},{tot['lds']},{tot['vmem']},{tot['salu']}")


if __name__ == "__main__":
    main()
