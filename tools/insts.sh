#!/bin/bash
# usage: tools/insts.sh lib...  -- SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS / time of the path kernel per experiment build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for l in "$@"; do
  rm -rf gpurun_out/insts_tmp
  PINE_GPU_LIB=pine_amd/lib/$l rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/insts_tmp -- python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
  echo -n "$l: "; python tools/pmc_summary.py gpurun_out/insts_tmp | grep -E "^SQ_INSTS_VALU|^SQ_INSTS_SALU|^SQ_INSTS_LDS|lane util|frac ACTIVE_INST_VALU" | awk '{printf "%s=%s  ", $1, $NF=="(n=3)"?$2:$NF} END{print ""}'
done
