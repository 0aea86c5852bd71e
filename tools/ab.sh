#!/bin/bash
# usage: tools/ab.sh lib1 lib2 ...   -- A/B experiment builds on ONE box: path_trace ms per lib, 2 rounds interleaved
for round in 1 2; do
for l in "$@"; do
  echo -n "$l: "; PINE_GPU_LIB=pine_amd/lib/$l python bench.py --steps 4 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernels_ms']['path_trace'],2), 'ms', round(d['value'],1), 'Ms/s')"
done; done
