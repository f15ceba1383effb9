#!/bin/bash
# round 6, call 19: batch-interleaved batch counts that are no multiple of a line (16 fp32 / 8 fp64 transforms): every 128-byte
# segment straddles two lines.  With / without the XCD-contiguous walk, register-resident and registered kernels.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_bi_unaligned.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 900 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_bi_unaligned_$tag.err; }
export PERF_STAGE_HX_CASES="f32:bi768@174768,f32:bi768@174762,f32:bi768@174769,f32:bi660@203360,f32:bi660@203365,f32:bi1024@131072,f32:bi1024@131077,f32:bi1024@132000,f32:bi1024@132005,f32:bi256@528000,f32:bi256@528005,f64:bi1024@66000,f64:bi1024@66003,f32:bi1000@134000,f32:bi1000@134007"
run contig PFFT_JIT_VERBOSE=0
run nocontig PFFT_XCD_CONTIG=0
unset PERF_STAGE_HX_CASES
cat $O
