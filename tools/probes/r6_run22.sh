#!/bin/bash
# round 6, call 22: four-step stages with an unaligned user side: default policy on that side only (experiment PFFT_FS_UNALIGNED=1)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_fs_unaligned.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 1200 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_fs_unaligned_$tag.err; }
export PERF_STAGE_HX_CASES="f32:68640,f32:1000000,f32:62500,f32:250000,f32:120000,f32:100000,f32:500000,f64:68640,f64:250000,f64:62500"
for rep in 1 2; do
run base PFFT_JIT_VERBOSE=0
run userside PFFT_FS_UNALIGNED=1
done
unset PERF_STAGE_HX_CASES
cat $O
