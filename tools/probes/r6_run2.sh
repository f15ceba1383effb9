#!/bin/bash
# round 6, call 2: register-resident strided stage kernels -- parity + time, against the LDS-resident twin, with and
# without the XCD-contiguous walk, the software-pipelined form, forced shapes for n = 1000; PMC traffic of ref68640
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_stage_hx.txt
: > $O
run() { tag=$1; shift; env "$@" timeout 900 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_stage_hx_$tag.err; }
run default PFFT_JIT_VERBOSE=0
run nohx PFFT_JIT_STRIDED_HX=0
run r5 PFFT_JIT_STRIDED_HX=0 PFFT_XCD_CONTIG=0
run hx_nocontig PFFT_XCD_CONTIG=0
run pf PFFT_JIT_STRIDED_HX=0 PFFT_JIT_STRIDED_PF=1
export PERF_STAGE_HX_CASES="f32:1000000,f32:bi1000,f32:2985984"
run f35x2 PFFT_JIT_STRIDED_HX_FORCE=35:2
run f20x2 PFFT_JIT_STRIDED_HX_FORCE=20:2
run f26x2 PFFT_JIT_STRIDED_HX_FORCE=26:2
unset PERF_STAGE_HX_CASES
cat $O
for c in ref68640; do python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline >> gpurun_out/r6_bench_hx.txt 2>&1; done
bash tools/run_pmc.sh ref68640 gpurun_out/r6_pmc_ref68640 > /dev/null 2>&1
grep -o '"frac": [0-9.]*' gpurun_out/r6_bench_hx.txt
