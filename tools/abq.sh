#!/bin/bash
# A/B of queue-kernel experiment builds on one box (2 interleaved rounds)
for round in 1 2; do
for l in "$@"; do
  echo -n "$l: "; PINE_GPU_KERNEL=queue PINE_GPU_LIB=pine_amd/lib/$l timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernels_ms']['path_trace'],2), 'ms', round(d['value'],1), 'Ms/s', d['config']['grid_blocks'])"
done; done
