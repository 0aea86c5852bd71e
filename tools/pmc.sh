#!/bin/bash
# usage: tools/pmc.sh <tag> [c2|c3|c4|c5]  -- collects SQ/TCC counters for bench.py's path kernel (one pass per group)
tag=$1
cfg=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --headline-only --config $cfg"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_A -- $B > gpurun_out/pmc_${tag}_A.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_B -- $B > gpurun_out/pmc_${tag}_B.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INST_LEVEL_VMEM --output-format csv -d gpurun_out/pmc_${tag}_C -- $B > gpurun_out/pmc_${tag}_C.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_D -- $B > gpurun_out/pmc_${tag}_D.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${tag}_E -- $B > gpurun_out/pmc_${tag}_E.log 2>&1
python tools/pmc_summary.py --traffic-json=gpurun_out/pmc_${tag}_traffic.json gpurun_out/pmc_${tag}_*/ > gpurun_out/pmc_${tag}_summary.txt 2>&1
cat gpurun_out/pmc_${tag}_summary.txt
