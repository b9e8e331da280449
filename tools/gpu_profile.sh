#!/bin/bash
# Kernel trace + three separate PMC passes of the bench frame; run ON the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/gpu_profile.sh r2a 192 f32'      (tag, samples per ray, precision)
# Writes gpurun_out/<tag>_S<samples>_<prec>/{trace,a,b,c}; summarise afterwards IN THE BUILD CONTAINER (where .git is) with
# tools/prof_summary.py / tools/pmc_summary.py, which stamp the commit.  Every rocprofv3 call is wrapped in `timeout` (an
# aborted counter run can otherwise hang to the box limit), and --pmc is only ever combined with --kernel-trace.
tag=${1:-prof}
S=${2:-192}
prec=${3:-f32}
repo=$(pwd)
export TMPDIR=/tmp
out=$repo/gpurun_out/${tag}_S${S}_${prec}
mkdir -p $out
cmd="python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --samples $S --precision $prec"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $cmd > $out/trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/a -o a -- $cmd > $out/a.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
    --kernel-trace --output-format csv -d $out/b -o b -- $cmd > $out/b.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $out/c -o c -- $cmd > $out/c.log 2>&1)
# keep what travels back small: the csv/db summaries only
find $out -name "*.json" -size +2M -delete 2>/dev/null
ls -R $out | head -40
