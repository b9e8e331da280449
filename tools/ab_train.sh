#!/bin/bash
# A/B timing of differently-built copies of the library (ab_*.so in the repo root, tools/ab_build.sh) on the TRAINING step:
#   gpurun -- 'bash tools/ab_train.sh [train_bench args]'   -> one line per library and sample count (S=192, then S=48)
# Each library is run twice per sample count, interleaved, so that a drift of the box shows up as a spread.
for rep in 1 2; do
  for so in ab_*.so; do
    for S in 192 48; do
      echo -n "$so rep $rep: "
      THERMONERF_HIP_LIB=$PWD/$so timeout 300 python tools/train_bench.py --samples $S --steps 120 --warmup 24 "$@" 2>/dev/null | tail -1
    done
  done
done
