#!/bin/bash
# round 6, call 11: experiments -- first factor of the two-stage BI plan, register-resident plans over registered fp32 entries (n = 512 x 32)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_exp11.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 600 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_exp11_$tag.err; }
export PERF_STAGE_HX_CASES="f32:bi2048@66000,f64:bi2048@33000"
run n1_default PFFT_JIT_VERBOSE=0
run n1_64 PFFT_BI_N1=64
run n1_16 PFFT_BI_N1=16
run n1_128 PFFT_BI_N1=128
export PERF_STAGE_HX_CASES="f32:bi4096@32768,f64:bi4096@16384,f32:bi4096@33000"
run n1_default PFFT_JIT_VERBOSE=0
run n1_32 PFFT_BI_N1=32
run n1_128 PFFT_BI_N1=128
run n1_256 PFFT_BI_N1=256
export PERF_STAGE_HX_CASES="f32:bi8192@16500,f32:bi16384@8200"
run n1_default PFFT_JIT_VERBOSE=0
run n1_64 PFFT_BI_N1=64
run n1_128 PFFT_BI_N1=128
run n1_256 PFFT_BI_N1=256
export PERF_STAGE_HX_CASES="f32:bi512@264000,f32:bi512@262144,f32:2d2048,f32:bi1024@132000"
run reg PFFT_JIT_VERBOSE=0
run hx32 PFFT_HX_OVER_REGISTERED=2
run reg PFFT_JIT_VERBOSE=0
run hx32 PFFT_HX_OVER_REGISTERED=2
unset PERF_STAGE_HX_CASES
cat $O
