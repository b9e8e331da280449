"""Summarise rocprofv3 --pmc csv passes into profiles/<name>.txt and refresh profiles/pmc_traffic.json.

usage: python tools/pmc_summary.py <dir with pass sub-dirs a/ b/ c/> <out.txt> <f32|f16x3> "<title>" [samples_per_ray=192]
(run in the build container on the csv files gpurun merged back; the summary is stamped with `git rev-parse HEAD` — commit
the kernels BEFORE the gpurun call that collects the counters, so the stamp is the commit that was measured)
Each pass dir holds *_counter_collection.csv written by
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir>/<pass> -o <pass> -- python bench.py ...
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are reported in KB, and on gfx950
FETCH_SIZE counts 64-byte units as 32 -> the fetched bytes are 2 x FETCH_SIZE KB.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KEEP = ("main_mfma_rays_kernel", "main_split_rays_kernel", "main_h3_rays_kernel", "proposal_rays_kernel", "main_mfma_kernel", "proposal_kernel")
KEY = {"main_mfma_rays_kernel": "field_render", "main_split_rays_kernel": "field_render", "main_h3_rays_kernel": "field_render",
       "proposal_rays_kernel": "proposal_sample"}


def short(name):
    for k in KEEP:
        if k + "(" in name or k + "<" in name:  # templated kernels print as name<args>(
            return k
    return None


def head_stamp():
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=root, capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--", "thermo_nerf_amd", "bench.py"], cwd=root,
                               capture_output=True, text=True, check=True).stdout.strip()
        return head + (" (+ uncommitted changes under thermo_nerf_amd/ or bench.py)" if dirty else "")
    except Exception:
        # on the GPU box there is no .git: the build container writes `git rev-parse HEAD` into tools/.head_stamp before the
        # gpurun call (tools/gpu_profile.sh header)
        try:
            return open(os.path.join(root, "tools", ".head_stamp")).read().strip()
        except OSError:
            return "unknown"


def main():
    d, dst, prec, title = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    S = int(sys.argv[5]) if len(sys.argv) > 5 else 192
    acc = defaultdict(list)  # (kernel, counter) -> per-dispatch values
    for p in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))):
        per = defaultdict(float)  # (dispatch, kernel, counter) -> summed over instances
        for row in csv.DictReader(open(p)):
            k = short(row["Kernel_Name"])
            if k:
                per[(row["Dispatch_Id"], k, row["Counter_Name"])] += float(row["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[(k, c)].append(v)
    with open(dst, "w") as out:
        out.write(f"# {title}\n# measured at commit {head_stamp()}\n")
        out.write("# separate --pmc passes (one sub-directory each); mean per dispatch.  GRBM_GUI_ACTIVE is summed over the 8 XCDs;\n"
                  "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles; FETCH_SIZE / WRITE_SIZE in KB as reported.\n")
        out.write("kernel,counter,dispatches,mean_per_dispatch\n")
        for (k, c), v in sorted(acc.items()):
            out.write(f"{k},{c},{len(v)},{sum(v) / len(v):.6g}\n")
    tj = os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_traffic.json")
    traffic = json.load(open(tj)) if os.path.exists(tj) else {}
    for k, name in KEY.items():
        f, w = acc.get((k, "FETCH_SIZE")), acc.get((k, "WRITE_SIZE"))
        if f and w:
            traffic[name + "@S%d" % S + ("" if prec == "f32" else "_" + prec)] = {
                "fetch_kb": 2.0 * sum(f) / len(f), "write_kb": sum(w) / len(w), "rays_per_launch": 640000,
                "commit": head_stamp(),
                "source": f"profiles/{os.path.basename(dst)} (2 x FETCH_SIZE per the MI355X_MICROARCH.md gfx950 note + "
                          "WRITE_SIZE, KB; 8-byte gathers, uncalibrated)"}
    json.dump(traffic, open(tj, "w"), indent=1)
    print(open(dst).read())


if __name__ == "__main__":
    main()
