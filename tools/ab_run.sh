#!/bin/bash
# A/B timing of differently-built copies of the library (ab_*.so in the repo root, see tools/ablate.py) on the bench frame:
#   gpurun -- 'bash tools/ab_run.sh [bench args]'     -> one line per library: frame ms, proposal ms, field ms
for so in ab_*.so; do
  THERMONERF_HIP_LIB=$PWD/$so python bench.py --no-variants --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{') and 'bench_detail' not in line[:20]:
        d=json.loads(line); r=d['roofline']
        print('$so', 'ms/frame %.3f' % d['ms_per_step'], 'proposal %.3f' % r['proposal_ms'], 'field %.3f' % r['field_ms'], 'frac %.4f' % r['frac'])
"
done
