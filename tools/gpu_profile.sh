#!/bin/bash
# Kernel trace + three separate PMC passes of the default bench frame, per precision; run ON the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/gpu_profile.sh v6'
# Writes gpurun_out/<tag>_<prec>/{trace,a,b,c}; summarise afterwards with tools/prof_summary.py / tools/pmc_summary.py.
# Every rocprofv3 call is wrapped in `timeout` (an aborted counter run can otherwise hang to the box limit), and
# --pmc is only ever combined with --kernel-trace.
tag=${1:-prof}
repo=$(pwd)
export TMPDIR=/tmp
for prec in f32 f16x3; do
  out=$repo/gpurun_out/${tag}_${prec}
  mkdir -p $out
  cmd="python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision $prec"
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $cmd > $out/trace.log 2>&1)
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $out/a -o a -- $cmd > $out/a.log 2>&1)
  (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE SQ_VALU_MFMA_COEXEC_CYCLES \
      --kernel-trace --output-format csv -d $out/b -o b -- $cmd > $out/b.log 2>&1)
  (cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum \
      --kernel-trace --output-format csv -d $out/c -o c -- $cmd > $out/c.log 2>&1)
done
ls -R $repo/gpurun_out/${tag}_f32 | head -30
