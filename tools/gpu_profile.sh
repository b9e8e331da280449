#!/bin/bash
# Kernel trace + three separate PMC passes of the bench frame, summarised ON the GPU box (the raw rocprofv3 databases exceed what
# gpurun copies back).  From the build container, on a COMMITTED tree:
#   git rev-parse HEAD > tools/.head_stamp && gpurun --timeout 1500 -- 'bash tools/gpu_profile.sh round2 192 f32'
# Writes gpurun_out/<tag>_kernel_trace_S<S>_<prec>.txt, gpurun_out/<tag>_pmc_S<S>_<prec>.txt and gpurun_out/pmc_traffic.json (pmc_summary.py
# writes the traffic entries NEXT TO the summary; tools/install_profiles.sh merges them into profiles/pmc_traffic.json — do not copy
# profiles/pmc_traffic.json over it).  Every rocprofv3 call is wrapped in `timeout`, and --pmc is only ever
# combined with --kernel-trace.
tag=${1:-prof}
S=${2:-192}
prec=${3:-f32}
repo=$(pwd)
export TMPDIR=/tmp
out=/tmp/prof_${tag}_S${S}_${prec}
rm -rf $out; mkdir -p $out $repo/gpurun_out
cmd="python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --samples $S --precision $prec"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $cmd > $out/trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/pmc/a -o a -- $cmd > $out/a.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
    --kernel-trace --output-format csv -d $out/pmc/b -o b -- $cmd > $out/b.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $out/pmc/c -o c -- $cmd > $out/c.log 2>&1)
cd $repo
python tools/prof_summary.py $out/trace gpurun_out/${tag}_kernel_trace_S${S}_${prec}.txt \
  "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --samples $S --precision $prec" > /dev/null
python tools/pmc_summary.py $out/pmc gpurun_out/${tag}_pmc_S${S}_${prec}.txt $prec \
  "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --samples $S --precision $prec (tools/gpu_profile.sh)" $S > /dev/null
tail -3 $out/trace.log
