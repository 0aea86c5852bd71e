#!/bin/bash
# usage: tools/sweep.sh  -- runs a few bench variants on the GPU box, prints value / path_trace ms
run() { echo -n "$1: "; env $2 python bench.py --steps 4 --warmup 1 --no-cpu $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['kernels_ms'], d['config']['grid_blocks'], d['config']['samples_per_item'])"; }
run "default" "" ""
L=PINE_GPU_LIB=pine_amd/lib/libpine_gpu_exp.so
run "wps4" "$L PINE_GPU_WPS=4" ""
run "wps5" "$L PINE_GPU_WPS=5" ""
run "wps6" "$L PINE_GPU_WPS=6" ""
run "wps8" "$L PINE_GPU_WPS=8" ""
