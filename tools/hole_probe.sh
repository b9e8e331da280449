#!/bin/bash
# Where does the gap before field_bwd_fused_kernel<1,8> come from?  Host API trace + kernel trace of a few training steps:
# for every kernel of the last steps: launch-call time (hipLaunchKernel / hipModuleLaunchKernel / hipExtModuleLaunchKernel enter),
# kernel begin / end; gap to the previous kernel's end on any stream, and how long before its begin the launch call returned.
S=${1:-192}
repo=$(pwd)
export TMPDIR=/tmp
out=/tmp/hole_S$S
rm -rf $out; mkdir -p $out $repo/gpurun_out
(cd /tmp && timeout 300 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $out -o h -- python $repo/tools/train_bench.py --steps 12 --warmup 6 --samples $S --ray-batch random > $out/log.txt 2>&1)
tail -2 $out/log.txt
python - "$out" <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
ht = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)
print("files", kt, ht)
K = list(csv.DictReader(open(kt[0])))
H = list(csv.DictReader(open(ht[0])))
print("kernel cols", list(K[0].keys()))
print("hip cols", list(H[0].keys()))
K.sort(key=lambda r: int(r["Start_Timestamp"]))
launches = [r for r in H if "Launch" in r["Function"]]
launches.sort(key=lambda r: int(r["Start_Timestamp"]))
by_corr = {r["Correlation_Id"]: r for r in launches}
# last ~2 steps
anchors = [i for i, r in enumerate(K) if "field_fwd_taped" in r["Kernel_Name"]]
a, b = anchors[-3], anchors[-2]
t0 = int(K[a]["Start_Timestamp"])
prev_end = None
for r in K[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    l = by_corr.get(r["Correlation_Id"])
    call = (int(l["Start_Timestamp"]) - t0) / 1e3 if l else float("nan")
    ret = (int(l["End_Timestamp"]) - t0) / 1e3 if l else float("nan")
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f %9.1f  gap %6.1f  launch call %9.1f..%9.1f (%+7.1f us before start)  %s" % (
        (s - t0) / 1e3, (e - t0) / 1e3, gap, call, ret, (s - t0) / 1e3 - ret, r["Kernel_Name"][:70]))
    prev_end = max(prev_end or e, e)
PY
