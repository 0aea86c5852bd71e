#!/bin/bash
# usage (on a one-GPU box): tools/rehearse_ranks.sh N [gather|reduce]
# Runs bench.py's N-rank code path with every rank on GPU 0 and the collectives staged through gloo/host
# memory: checks the multi-rank logic (shards, slabs, gather/unpack or reduce, JSON) end to end; the
# film md5 it prints must equal the 1-rank md5.  Timings of such a run mean nothing.
N=${1:-2}
export PINE_BENCH_BACKEND=gloo PINE_BENCH_DEVICE=0 PINE_BENCH_COLLECTIVE=${2:-gather}
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ranks', d['n_gpus'], d['config']['collective'], 'film_md5', d['film_md5'], 'vertices/sample', d['config']['vertices_per_sample'])"
