#!/bin/bash
# Per-kernel durations of the training step for differently-built copies of the library (ab_*.so, tools/ab_build.sh):
#   gpurun -- 'bash tools/ab_train_kernels.sh <samples> [kernel name pattern] [extra train_bench args]'
# runs the bucketed-scatter parity tests and a rocprofv3 kernel trace of tools/train_bench.py with each library and prints the
# step time + the matching kernels' calls / average duration.
S=${1:-192}
pat=${2:-"sort_emit|sort_owner|hash_encode_bwd_kernel"}
shift; shift
for so in ab_*.so; do
  tag=$(basename $so .so)
  export THERMONERF_HIP_LIB=$PWD/$so
  echo "== $so"
  timeout 600 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "bucketed" 2>&1 | tail -1
  bash tools/train_profile.sh $tag $S "$@" | tail -1
  grep -E "$pat" gpurun_out/${tag}_kernel_trace_train_S${S}.txt | sed 's/(anonymous namespace):://g' | awk -F'",' '{split($2,a,","); n=$1; sub(/^"/,"",n); sub(/\(.*/,"",n); printf "   %-60s calls %s avg %s us\n", n, a[1], a[3]}'
done
