#!/bin/bash
# usage (GPU box): tools/ab.sh "<configs>" lib [lib ...]   e.g. tools/ab.sh "c2 c4 c5" libpine_gpu.so libpine_gpu_x.so
# Path-kernel time (HIP events) and film md5 vs the reference's of each BASELINE config on each built library.
set -o pipefail
cfgs=$1; shift
for c in $cfgs; do
  for l in "$@"; do
    steps=6; [ $c = c5 ] && steps=3
    PINE_GPU_LIB=pine_amd/lib/$l timeout -k 10 300 python bench.py --config $c --steps $steps --warmup 1 --no-cpu --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('%-28s %s' % ('$l', '$c'), 'ms_per_step', round(j['ms_per_step'],3), 'Ms/s', round(j['value'],1), 'eq_ref', j.get('film_equals_reference'), 'kernel_ms', round(j['kernels_ms']['path_trace'],3), 'frac', round(j['roofline']['frac'],4))
" || echo "$l $c FAILED"
  done
done
