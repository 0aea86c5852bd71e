#!/bin/bash
# usage (GPU box): tools/flag_sweep.sh [config]  -- the scene-specialised kernel compiled with extra compiler flags ($PINE_GPU_SPECIALIZE_EXTRA)
cfg=${1:-c2}
mkdir -p gpurun_out/r3
run() {
  PINE_GPU_SPECIALIZE_EXTRA="$1" timeout -k 10 200 python bench.py --config $cfg --steps 8 --warmup 2 --headline-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('%-62s' % sys.argv[1], '$cfg kernel_ms', round(j['kernels_ms']['path_trace'],3), 'eq_ref', j.get('film_equals_reference'), j['config']['kernel'][:18])
" "$1"
}
run ""
run "-O2"
run "-mllvm -amdgpu-sched-strategy=max-ilp"
run "-mllvm -amdgpu-sched-strategy=max-memory-clause"
run "-mllvm -amdgpu-sched-strategy=iterative-ilp"
run "-mllvm -amdgpu-sched-strategy=iterative-minreg"
run "-mllvm -enable-post-misched=0"
run "-fno-unroll-loops"
run "-mllvm -amdgpu-early-inline-all=true"
run "-mllvm -misched-cluster=0"
run "-mllvm -amdgpu-use-aa-in-codegen=1"
run "-mllvm -enable-misched=0"
run "-mllvm -amdgpu-schedule-relaxed-occupancy=1"
run ""
