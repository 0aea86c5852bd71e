#!/bin/bash
# usage (GPU box): tools/c4_x.sh  -- C4 path-kernel time: flat traversal inside stages S / T (default) against traversal STAGES with
# lane refill, both with the scene's exact feature set (the stage variant through the run-time compiler: PINE_GPU_XSTAGE=1 picks
# the all-but-Subsurface stage variant, whose layout the scene's own kernel inherits)
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" python3 bench.py --headline-only --config ${CFG:-c4} --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['roofline']['kernel'], d['roofline']['kernel_ms'], 'ms', d['config']['kernel_mode'], d['film_equals_reference'])"
}
run X=0
run PINE_GPU_XSTAGE=1 PINE_GPU_TRAV_MIN_LANES=0
for l in 32 48 56; do for t in 2 4 8; do run PINE_GPU_XSTAGE=1 PINE_GPU_TRAV_MIN_LANES=$l PINE_GPU_TRAV_MIN_TRIPS=$t; done; done
