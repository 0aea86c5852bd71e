#!/bin/bash
# usage: tools/pmc.sh <tag> [c2|c3|c4|c5] [specialised|precompiled] [counters.json]
# Collects SQ / TCC counters for bench.py's path kernel: one rocprofv3 --pmc pass per counter group, --kernel-trace only (no
# other trace domain beside --pmc).  The program after `--` is python itself.  The kernel cache is warmed by a plain run first,
# so that no compiler child is ever started from a profiled process (its environment would be scrubbed anyway).
tag=$1
cfg=${2:-c2}
mode=${3:-specialised}
counters=${4:-gpurun_out/${tag}_counters.json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$mode" = precompiled ]; then export PINE_BENCH_SPECIALIZE=0; else unset PINE_BENCH_SPECIALIZE; fi
python3 bench.py --steps 1 --warmup 1 --headline-only --config $cfg > gpurun_out/pmc_${tag}_warm.log 2>&1
B="python3 bench.py --steps 2 --warmup 1 --headline-only --config $cfg"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_A -- $B > gpurun_out/pmc_${tag}_A.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_B -- $B > gpurun_out/pmc_${tag}_B.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INST_LEVEL_VMEM --output-format csv -d gpurun_out/pmc_${tag}_C -- $B > gpurun_out/pmc_${tag}_C.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_D -- $B > gpurun_out/pmc_${tag}_D.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${tag}_E -- $B > gpurun_out/pmc_${tag}_E.log 2>&1
python3 tools/pmc_summary.py --traffic-json=gpurun_out/pmc_${tag}_traffic.json --counters-json=$counters --config=$cfg --mode=$mode gpurun_out/pmc_${tag}_[A-E]/ > gpurun_out/pmc_${tag}_summary.txt 2>&1
cat gpurun_out/pmc_${tag}_summary.txt
