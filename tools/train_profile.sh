#!/bin/bash
# rocprofv3 kernel trace of the training step; run ON the GPU box:  gpurun -- 'bash tools/train_profile.sh <tag> [samples]'
tag=${1:-train}
S=${2:-48}
repo=$(pwd)
export TMPDIR=/tmp
out=$repo/gpurun_out/${tag}_S${S}
mkdir -p $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $repo/tools/train_bench.py --steps 36 --warmup 6 --samples $S > $out/trace.log 2>&1)
tail -2 $out/trace.log
find $out -name "*.json" -size +2M -delete 2>/dev/null
ls $out/trace | head
