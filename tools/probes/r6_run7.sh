#!/bin/bash
# round 6, call 7: experiments -- register-resident stage A in registered pairs (C3 and friends), chunk overlap for plans
# whose stage A is compiled at commit
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_exp7.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 900 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_exp7_$tag.err; }
export PERF_STAGE_HX_CASES="f64:1048576,f64:2097152,f64:524288,f32:1048576,f32:2097152,f64:3145728"
for rep in 1 2; do
run base PFFT_JIT_VERBOSE=0
run hxA PFFT_HX_STAGE_A=1
run hxA2 PFFT_HX_STAGE_A=2
done
export PERF_STAGE_HX_CASES="f32:68640,f32:1000000,f32:62500,f32:250000,f32:120000,f32:2985984,f64:68640,f64:1000000,f64:250000,f64:120000"
for rep in 1 2; do
run base PFFT_JIT_VERBOSE=0
run anyorder PFFT_JIT_ANY_ORDER=1
done
unset PERF_STAGE_HX_CASES
cat $O
for e in "" "PFFT_HX_STAGE_A=1"; do for rep in 1 2; do env $e python bench.py --config c3 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('c3 [$e]', d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_wall'])"; done; done
