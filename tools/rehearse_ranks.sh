#!/bin/bash
# usage (on a one-GPU box): tools/rehearse_ranks.sh N [gather|reduce]
# Runs bench.py's N-rank code path THROUGH ITS OWN LAUNCHER with every rank on GPU 0 and the collectives staged through
# gloo/host memory: checks the multi-rank logic (ranks started by `bench.py --gpus N`, shards, slabs, gather/unpack or reduce,
# the line) end to end; films must equal the reference's md5 (C2 headline and C3 on the same ranks).  Timings of such a run mean nothing.
N=${1:-2}
export PINE_BENCH_BACKEND=gloo PINE_BENCH_DEVICE=0 PINE_BENCH_COLLECTIVE=${2:-gather}
python3 bench.py --gpus $N --steps 2 --warmup 1 --no-cpu --detail gpurun_out/rehearse_${N}_detail.json 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ranks', d['n_gpus'], d['config']['collective'], d['config']['kernel_mode'], 'C2 film == reference:', d['film_equals_reference'], '| side configs:', [(c['name'], c['ok']) for c in d.get('configs', [])])"
