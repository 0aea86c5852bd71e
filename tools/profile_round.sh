#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh rNN
# Produces profiles/<tag>_* : rocprofv3 kernel-trace stats of `python bench.py`, PMC passes of the
# path kernel (separate passes, no trace domains besides kernel-trace), and the bench JSON line.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
# (--no-configs: only the headline workload's launches, so that the per-kernel AVERAGE is C2's -- C3 runs the same kernel variant)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 10 --warmup 2 --no-configs --no-cpu > gpurun_out/profiles/${tag}_bench_under_rocprof.log 2>&1
cp gpurun_out/prof_$tag/*/*_kernel_stats.csv gpurun_out/profiles/${tag}_kernel_stats.csv
tools/pmc.sh $tag > /dev/null 2>&1
cp gpurun_out/pmc_${tag}_summary.txt gpurun_out/profiles/${tag}_pmc_summary.txt
cp gpurun_out/pmc_${tag}_traffic.json gpurun_out/profiles/${tag}_traffic.json
mkdir -p profiles && cp gpurun_out/pmc_${tag}_traffic.json profiles/${tag}_traffic.json  # so that the bench line below carries it
# the other configs' path kernels: kernel-trace stats + counters + fabric traffic each
for c in c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$c -- python bench.py --steps 3 --warmup 1 --no-cpu --no-configs --config $c > gpurun_out/profiles/${tag}_${c}_under_rocprof.log 2>&1
  cp gpurun_out/prof_${tag}_$c/*/*_kernel_stats.csv gpurun_out/profiles/${tag}_${c}_kernel_stats.csv
  tools/pmc.sh ${tag}_$c $c > /dev/null 2>&1
  cp gpurun_out/pmc_${tag}_${c}_summary.txt gpurun_out/profiles/${tag}_${c}_pmc_summary.txt
  cp gpurun_out/pmc_${tag}_${c}_traffic.json gpurun_out/profiles/${tag}_traffic_${c}.json
  cp gpurun_out/pmc_${tag}_${c}_traffic.json profiles/${tag}_traffic_${c}.json
done
# (the bench line reads its VALU figures from the latest profiles/rNN_[cX_]pmc_summary.txt: this round's)
cp gpurun_out/profiles/${tag}_pmc_summary.txt gpurun_out/profiles/${tag}_c4_pmc_summary.txt gpurun_out/profiles/${tag}_c5_pmc_summary.txt profiles/ 2>/dev/null
python bench.py --steps 10 --warmup 2 > gpurun_out/profiles/${tag}_bench.json 2> gpurun_out/profiles/${tag}_bench.err
tail -1 gpurun_out/profiles/${tag}_bench.json | cut -c1-400
cat gpurun_out/profiles/${tag}_kernel_stats.csv | cut -c1-160
