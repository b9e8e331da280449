#!/bin/bash
# A/B of round 6's deferred table update on sustained runs (random-pixel batches).  Usage: tools/ab_deferred.sh [seconds] [S ...]
SEC=${1:-6}; shift
SS=${@:-192 48}
for S in $SS; do
  for mode in "" "--per-call" "--joined-table" "--per-call --joined-table" "--per-call --torch-adam"; do
    echo "== S=$S ${mode:-deferred (default)}"
    python tools/train_bench.py --samples $S --ray-batch random --seconds $SEC $mode 2>&1 | tail -1
  done
done
