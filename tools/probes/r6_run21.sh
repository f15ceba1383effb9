#!/bin/bash
# round 6, call 21: default cache policies + shared walk for stages with unaligned row pitches (PFFT_NO_UNALIGNED_POLICY=1 is the twin):
# batch-interleaved at odd batch counts (registered and runtime lengths, one- and two-stage plans), the runtime four-step lengths
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_unaligned_policy.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 1200 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_unaligned_policy_$tag.err; }
export PERF_STAGE_HX_CASES="f32:bi768@174769,f32:bi1024@131077,f32:bi1024@33000,f32:bi256@528005,f32:bi64@2000003,f64:bi1024@66003,f32:bi1000@134007,f32:bi4096@33001,f32:bi2048@66003,f32:bi1024@131072,f32:68640,f32:1000000,f32:62500,f32:250000,f32:120000,f32:2985984,f32:100000,f32:500000,f64:68640,f64:1000000,f64:250000,f64:62500,f32:2d1000"
for rep in 1 2; do
run new PFFT_JIT_VERBOSE=0
run old PFFT_NO_UNALIGNED_POLICY=1
done
unset PERF_STAGE_HX_CASES
cat $O
