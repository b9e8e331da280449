#!/bin/bash
# A/B on one box: mlp_base's output rows kept by the forward (round 5) against recomputed in the backward's head launches
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for S in 192 48; do
  for args in "--no-store-base" ""; do
    echo "S=$S ${args:-store-base}: $(python tools/train_bench.py --steps 360 --warmup 40 --samples $S --ray-batch random $args 2>&1 | tail -1)"
  done
done
done
