#!/bin/bash
# usage (GPU box): CFG=c4 tools/insts_env.sh "VAR=1 VAR2=x" "..."   -- VALU / SALU instruction counts, lane utilisation and time of the
# path kernel of `bench.py --headline-only --config $CFG` under each environment (one rocprofv3 --pmc pass each; the kernel cache is
# warmed by a plain run first)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in "$@"; do
  rm -rf gpurun_out/insts_tmp
  env $e python3 bench.py --steps 1 --warmup 1 --headline-only --config ${CFG:-c4} > /dev/null 2>&1
  env $e rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/insts_tmp -- python3 bench.py --steps 2 --warmup 1 --headline-only --config ${CFG:-c4} > gpurun_out/insts_tmp.log 2>&1
  echo -n "$e: "; python3 tools/pmc_summary.py gpurun_out/insts_tmp | grep -E "^SQ_INSTS_VALU |^SQ_INSTS_SALU|^SQ_INSTS_LDS|^SQ_INSTS_VMEM_RD|^SQ_WAVE_CYCLES|lane util" | awk '{printf "%s=%s  ", $1, ($1=="VALU")?$NF:$2} END{print ""}'
done
