#!/bin/bash
# usage (GPU box): tools/sweep.sh LIB CONFIG VAR v1 v2 ...   -- path-kernel ms of one config with an environment knob at each value
lib=$1; cfg=$2; var=$3; shift 3
steps=5; [ $cfg = c5 ] && steps=2
for v in "$@"; do
  env $var=$v PINE_GPU_LIB=pine_amd/lib/$lib timeout -k 10 300 python bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('$cfg $var=$v', 'kernel_ms', round(j['kernels_ms']['path_trace'],3), 'eq_ref', j.get('film_equals_reference'))
" || echo "$cfg $var=$v FAILED"
done
