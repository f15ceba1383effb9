#!/bin/bash
# round 6, call 20: unaligned batch-interleaved batches -- store / load cache policies of the kernels compiled at commit
# (PFFT_JIT_NT_AUX: 2 = nt both (default), 0x102 = nt loads + write-back stores, 0 = default both, 0x300 = default loads + nt stores)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_bi_unaligned_policies.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 900 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_bi_unaligned_pol_$tag.err; }
export PERF_STAGE_HX_CASES="f32:bi768@174768,f32:bi768@174769,f32:bi660@203365,f32:bi1000@134000,f32:bi1000@134007,f64:bi660@101683,f32:bi300@447397"
run nt PFFT_JIT_VERBOSE=0
run wb_stores PFFT_JIT_NT_AUX=0x102
run all_default PFFT_JIT_NT_AUX=0
run nt_stores_only PFFT_JIT_NT_AUX=0x300
run wb_nocontig PFFT_JIT_NT_AUX=0x102 PFFT_XCD_CONTIG=0
unset PERF_STAGE_HX_CASES
cat $O
