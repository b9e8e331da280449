#!/bin/bash
# deferred table update (the Trainer's default) against the joined form, alternating runs.  Usage: tools/ab_joined.sh [seconds] [reps] [S ...]
SEC=${1:-5}; REPS=${2:-3}; shift; shift
SS=${@:-192 48}
for S in $SS; do
  for rep in $(seq 1 $REPS); do
    for mode in "" "--joined-table"; do
      echo "== S=$S ${mode:-deferred} run $rep"
      python tools/train_bench.py --samples $S --ray-batch random --seconds $SEC $mode 2>&1 | tail -1
    done
  done
done
