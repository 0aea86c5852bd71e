#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh rNN
# Produces gpurun_out/profiles/<tag>_* (copy what is to be judged into profiles/): rocprofv3 kernel-trace stats of
# `python bench.py`, PMC passes of the path kernel per config and kernel mode (separate passes, no trace domain besides
# kernel-trace) merged into <tag>_counters.json, and the bench line + detail file of a plain run that reads those counters.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles
mkdir -p $out profiles
counters=$out/${tag}_counters.json
rm -f $counters
# warm the kernel cache in plain runs (no profiler attached while a compiler child runs)
for c in c2 c5; do python3 bench.py --steps 1 --warmup 1 --headline-only --config $c > $out/${tag}_warm_$c.log 2>&1; done
# kernel-trace stats: the default-mode headline (C2; --headline-only: every path-kernel launch of the run is one of the headline
# loop's -- 3 warm-up + 20 timed -- so the per-kernel AVERAGE is comparable with the HIP-event mean the bench line reports) ...
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 20 --warmup 3 --headline-only > $out/${tag}_bench_under_rocprof.log 2>&1
cp gpurun_out/prof_$tag/*/*_kernel_stats.csv $out/${tag}_kernel_stats.csv
echo "progress: c2 trace done"
tools/pmc.sh $tag c2 specialised $counters > /dev/null 2>&1
cp gpurun_out/pmc_${tag}_summary.txt $out/${tag}_pmc_summary.txt
tools/pmc.sh ${tag}_c2pre c2 precompiled $counters > /dev/null 2>&1
cp gpurun_out/pmc_${tag}_c2pre_summary.txt $out/${tag}_c2_precompiled_pmc_summary.txt
echo "progress: c2 counters done"
# ... and the other configs' path kernels
for c in c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$c -- python3 bench.py --steps 6 --warmup 2 --headline-only --config $c > $out/${tag}_${c}_under_rocprof.log 2>&1
  cp gpurun_out/prof_${tag}_$c/*/*_kernel_stats.csv $out/${tag}_${c}_kernel_stats.csv
  mode=specialised; [ $c = c4 ] && mode=precompiled   # (C4 has nothing to specialise: its variant is its feature set)
  tools/pmc.sh ${tag}_$c $c $mode $counters > /dev/null 2>&1
  cp gpurun_out/pmc_${tag}_${c}_summary.txt $out/${tag}_${c}_pmc_summary.txt
  echo "progress: $c done"
done
tools/pmc.sh ${tag}_c5pre c5 precompiled $counters > /dev/null 2>&1
cp gpurun_out/pmc_${tag}_c5pre_summary.txt $out/${tag}_c5_precompiled_pmc_summary.txt
cp $counters profiles/${tag}_counters.json   # so that the bench line below carries this round's counters
python3 bench.py --steps 10 --warmup 2 --detail $out/${tag}_bench_detail.json > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json | cut -c1-600
cut -c1-160 $out/${tag}_kernel_stats.csv
