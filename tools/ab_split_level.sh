#!/bin/bash
# A/B of the table scatter's split level (atomics below it, bucketed records from it on) under the deferred table update:
# sustained runs on random-pixel batches.  Usage: tools/ab_split_level.sh [seconds] [S ...]
SEC=${1:-6}; shift
SS=${@:-192 48}
for S in $SS; do
  for lvl in -1 7 8 9 10; do
    for rep in 1 2; do
      echo "== S=$S first sorted level $lvl (-1 = the library's choice) run $rep"
      python tools/train_bench.py --samples $S --ray-batch random --seconds $SEC --first-sorted-level $lvl 2>&1 | tail -1
    done
  done
done
