#!/bin/bash
# round 6, call 15: batch-interleaved arrays of 4 GiB and more on the BIG forms (parity + time, against PFFT_NO_BIG_BI=1),
# the pre-compiled pow2 four-step plans back on the loop of rounds 1-5 (g32_20 ... g32_24), strided parity tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in g32_20 g32_21 g32_22 g32_24 g32_19 c3; do python bench.py --config $c --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$c', d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_wall'])"; done | tee gpurun_out/r6_pow2_recheck.txt
O=gpurun_out/r6_big_bi.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 1200 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_big_bi_$tag.err; }
export PERF_STAGE_HX_CASES="f32:bi4096@131008,f32:bi4096@131072,f32:bi4096@131136,f32:bi2048@300000,f64:bi4096@70000,f32:bi8192@70000,f32:bi4096@250000"
run big PFFT_JIT_VERBOSE=0
run nobig PFFT_NO_BIG_BI=1
unset PERF_STAGE_HX_CASES
cat $O; tail -3 gpurun_out/r6_big_bi_big.err
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "strided or register_resident_stage or global_sizes or runtime_specialised" 2>&1 | tail -3
