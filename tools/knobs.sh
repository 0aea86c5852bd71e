#!/bin/bash
# usage (GPU box): tools/knobs.sh CONFIG "ENV=VAL ENV=VAL" "..."   -- path-kernel time of one bench config per environment setting
cfg=$1; shift
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('$cfg [$e]', 'trace_ms', round(j['kernels_ms']['path_trace'],3), 'Ms/s', round(j['value'],1), 'eq_ref', j.get('film_equals_reference'))
"
done
