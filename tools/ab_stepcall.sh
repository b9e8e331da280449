#!/bin/bash
# host-side A/B of round 6's step calls at the reference's default S = 48 (and S = 192): per 100-step window the device backlog and
# the host's ms per phase (forward / losses / backward / optimizer).  Usage: tools/ab_stepcall.sh [seconds] [S ...]
SEC=${1:-6}; shift
SS=${@:-48}
for S in $SS; do
  for rep in 1 2; do
  for mode in "" "--joined-table" "--per-call" "--per-call --joined-table"; do
    echo "== S=$S rep $rep ${mode:-step calls + deferred table update (default)}"
    python tools/train_bench.py --samples $S --ray-batch random --seconds $SEC $mode 2>&1 | tail -2 | cut -c1-1500
  done
  done
done
