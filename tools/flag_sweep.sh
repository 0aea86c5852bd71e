#!/bin/bash
# usage (GPU box): tools/flag_sweep.sh [config]  -- the scene's own kernel compiled with extra compiler flags, one run-time compile
# per row (tools/spec_extra_ab.sh; PINE_GPU_SPECIALIZE_FORCE makes configs with nothing to specialise compile their kernel too)
CFG=${1:-c2} tools/spec_extra_ab.sh "" "-O2" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" \
  "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=iterative-minreg" "-mllvm -enable-post-misched=0" \
  "-mllvm -amdgpu-early-inline-all=true" "-mllvm -misched-cluster=0" "-mllvm -amdgpu-use-aa-in-codegen=1" "-mllvm -enable-misched=0" \
  "-mllvm -amdgpu-schedule-relaxed-occupancy=1" "-mllvm -amdgpu-early-ifcvt=1" "-mllvm -two-entry-phi-node-folding-threshold=16" \
  "-mllvm -phi-node-folding-threshold=8" "-mllvm -amdgpu-skip-threshold=32" "-mllvm -amdgpu-skip-threshold=4" \
  "-mllvm -simplifycfg-merge-cond-stores=true -mllvm -speculate-one-expensive-inst=true" ""
