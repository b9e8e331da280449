#!/bin/bash
# rocprofv3 kernel trace of the training step, summarised on the GPU box:
#   git rev-parse HEAD > tools/.head_stamp && gpurun -- 'bash tools/train_profile.sh <tag> [samples] [extra train_bench args]'
tag=${1:-train}
S=${2:-48}
shift; shift
extra="$*"
repo=$(pwd)
export TMPDIR=/tmp
out=/tmp/prof_${tag}_S${S}
rm -rf $out; mkdir -p $out $repo/gpurun_out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $repo/tools/train_bench.py --steps 36 --warmup 6 --samples $S $extra > $out/trace.log 2>&1)
cd $repo
python tools/prof_summary.py $out/trace gpurun_out/${tag}_kernel_trace_train_S${S}.txt \
  "rocprofv3 --kernel-trace --stats -- python tools/train_bench.py --steps 36 --warmup 6 --samples $S $extra (42 steps in the trace)" > /dev/null
python tools/step_timeline.py $out/trace gpurun_out/${tag}_timeline_train_S${S}.txt > /dev/null 2>gpurun_out/${tag}_timeline_err.txt || true
python tools/step_timeline.py $out/trace gpurun_out/${tag}_timeline_train_S${S}_update_step.txt field_fwd_taped density_bwd_train > /dev/null 2>>gpurun_out/${tag}_timeline_err.txt || true
grep "ms/step" $out/trace.log
