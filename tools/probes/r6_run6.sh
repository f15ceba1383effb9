#!/bin/bash
# round 6, call 6: scattered group order at power-of-two pitches (experiment), the re-measured splits at 1 GiB against their
# round-5 entries (PFFT_GLOBAL_N1 = the old first factor)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_exp6.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 600 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_exp6_$tag.err; }
export PERF_STAGE_HX_CASES="f32:bi1024@524288,f64:bi1024@262144,f32:bi256@2097152,f32:bi1024@400000,f32:bi4096@32768,f32:bi768@262144,f32:2d1024,f64:2d1024,f32:1048576,f64:1048576"
run plain PFFT_JIT_VERBOSE=0
run scatter PFFT_GROUP_SCATTER=1
run plain2 PFFT_JIT_VERBOSE=0
run scatter2 PFFT_GROUP_SCATTER=1
for pair in f32:250000:200 f32:200000:200 f64:120000:100 f64:200000:500 f64:1000000:1000; do
  IFS=: read prec n old <<< "$pair"
  export PERF_STAGE_HX_CASES="$prec:$n"
  run new_$n PFFT_JIT_VERBOSE=0
  run old_$n PFFT_GLOBAL_N1=$old
  run new_$n PFFT_JIT_VERBOSE=0
  run old_$n PFFT_GLOBAL_N1=$old
done
unset PERF_STAGE_HX_CASES
cat $O
