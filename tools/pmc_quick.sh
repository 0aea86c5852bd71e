#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <config>  -- two PMC passes (VALU / wait shares, instruction mix) of bench.py's path kernel
tag=$1
cfg=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu --no-configs --config $cfg"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_A -- $B > gpurun_out/pmc_${tag}_A.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_B -- $B > gpurun_out/pmc_${tag}_B.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${tag}_*/ 2>&1 | grep -E "kernel|frac|lane|VALU insts"
grep -o '"kernel_ms": [0-9.]*' gpurun_out/pmc_${tag}_A.log | head -1
