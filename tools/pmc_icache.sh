#!/bin/bash
# usage: tools/pmc_icache.sh lib  -- instruction-cache and wait-state counters of the path kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
l=$1
rm -rf gpurun_out/ic_tmp
PINE_GPU_LIB=pine_amd/lib/$l rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d gpurun_out/ic_tmp -- python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/ic_tmp
rm -rf gpurun_out/ic_tmp2
PINE_GPU_LIB=pine_amd/lib/$l rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/ic_tmp2 -- python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/ic_tmp2
