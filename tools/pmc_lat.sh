#!/bin/bash
# usage: tools/pmc_lat.sh lib  -- in-flight level counters (latency x count) for LDS / VMEM / SMEM of the path kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
l=$1
rm -rf gpurun_out/lat_tmp
PINE_GPU_LIB=pine_amd/lib/$l rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM --output-format csv -d gpurun_out/lat_tmp -- python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/lat_tmp | grep "^SQ_"
