#!/bin/bash
# usage (GPU box): tools/ab.sh name1 name2 ...   -- bench each pine_amd/lib/x_<name>.so (A/B experiments)
for n in "$@"; do
  PINE_GPU_LIB=pine_amd/lib/x_$n.so timeout -k 10 120 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('$n', 'ms_per_step', round(j['ms_per_step'],3), 'Ms/s', round(j['value'],1), 'md5', j.get('film_md5'))
"
done
