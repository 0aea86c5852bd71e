#!/bin/bash
# usage (GPU box): tools/quick_ab.sh "<configs>" "<ENV=... settings A>" "<settings B>" ...   -- bench.py --headline-only under each environment
set -o pipefail
cfgs=$1; shift
for c in $cfgs; do
  for e in "$@"; do
    steps=6; [ $c = c5 ] && steps=3
    env $e timeout -k 10 300 python bench.py --config $c --steps $steps --warmup 1 --headline-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('%-60s %s' % ('$e', '$c'), 'ms_per_step', round(j['ms_per_step'],3), 'Ms/s', round(j['value'],1), 'eq_ref', j.get('film_equals_reference'), 'kernel_ms', round(j['kernels_ms']['path_trace'],3), 'frac', round(j['roofline']['frac'],4))
" || echo "$e $c FAILED"
  done
done
