#!/bin/bash
# PMC passes of the training step (rocprofv3 --pmc with --kernel-trace only), summarised on the GPU box:
#   git rev-parse HEAD > tools/.head_stamp && gpurun -- 'bash tools/train_pmc.sh <tag> [samples] [kernel regex] [extra train_bench args]'
# Writes gpurun_out/<tag>_pmc_train_S<S>.txt: mean counter value per dispatch for every kernel whose name matches the regex.
tag=${1:-train}
S=${2:-192}
rx=${3:-field_bwd_fused_kernel|field_fwd_taped_kernel|hash_encode_bwd|sort_|linear_chain}
repo=$(pwd)
export TMPDIR=/tmp
out=/tmp/pmc_${tag}_S${S}
rm -rf $out; mkdir -p $out $repo/gpurun_out
shift; shift; shift
cmd="python $repo/tools/train_bench.py --steps 12 --warmup 3 --samples $S $*"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/a -o a -- $cmd > $out/a.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d $out/b -o b -- $cmd > $out/b.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT \
    --kernel-trace --output-format csv -d $out/c -o c -- $cmd > $out/c.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum \
    --kernel-trace --output-format csv -d $out/d -o d -- $cmd > $out/d.log 2>&1)
cd $repo
python - "$out" "gpurun_out/${tag}_pmc_train_S${S}.txt" "$rx" "$S" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
d, dst, rx, S = sys.argv[1:5]
rx = re.compile(rx)
acc = defaultdict(list)
for p in sorted(glob.glob(os.path.join(d, "*", "**", "*_counter_collection.csv"), recursive=True)):
    per = defaultdict(float)
    for row in csv.DictReader(open(p)):
        name = row["Kernel_Name"]
        if rx.search(name):
            k = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
            per[(row["Dispatch_Id"], k, row["Counter_Name"])] += float(row["Counter_Value"])
    for (_, k, c), v in per.items():
        acc[(k, c)].append(v)
try:
    stamp = open("tools/.head_stamp").read().strip()
except OSError:
    stamp = "unknown"
with open(dst, "w") as out:
    out.write(f"# rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python tools/train_bench.py --steps 12 --warmup 3 --samples {S} (tools/train_pmc.sh)\n")
    out.write(f"# measured at commit {stamp}\n# separate --pmc passes; mean per dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles; FETCH_SIZE / WRITE_SIZE in KB as reported\n")
    out.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (k, c), v in sorted(acc.items()):
        out.write(f"{k},{c},{len(v)},{sum(v) / len(v):.6g}\n")
print(open(dst).read())
PY
for f in a b c d; do tail -1 $out/$f.log; done
