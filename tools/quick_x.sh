#!/bin/bash
# usage (GPU box): tools/quick_x.sh [lib ...]  -- parity of three X-variant scenes on the current library, then C4 / C5 bench per library
set -o pipefail
mkdir -p gpurun_out/r2
for s in classic12 sss classic20; do timeout -k 10 120 python tools/one_scene.py $s 2>&1 | grep -v amdgpu.ids || exit 1; done
for l in "$@"; do
  for c in c4 c5; do
    PINE_GPU_LIB=pine_amd/lib/$l timeout -k 10 200 python bench.py --config $c --steps 4 --warmup 1 --no-cpu --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('$l $c', 'ms_per_step', round(j['ms_per_step'],3), 'Ms/s', round(j['value'],1), 'eq_ref', j.get('film_equals_reference'), 'trace_ms', j['kernels_ms']['path_trace'])
"
  done
done
