#!/bin/bash
# usage (GPU box): CFG=c2 tools/spec_extra_ab.sh "" "-DFOO" "-DBAR=2" ...   -- path-kernel time of the scene-specialised kernel of one
# bench config compiled with extra flags (one run-time compile each through PINE_GPU_SPECIALIZE_EXTRA: no library rebuild);
# PINE_GPU_SPECIALIZE_FORCE=1 so that configs with nothing to specialise (C4) compile their feature-set kernel too.
cd $GRAFT_REPO_ROOT
for x in "$@"; do
  echo -n "[$x] : "
  PINE_GPU_SPECIALIZE=1 PINE_GPU_SPECIALIZE_FORCE=1 PINE_GPU_SPECIALIZE_EXTRA="$x" python3 bench.py --headline-only --config ${CFG:-c2} --steps ${STEPS:-5} --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['roofline']['kernel'], d['roofline']['kernel_ms'], 'ms  film ok:', d['film_equals_reference'])"
done
