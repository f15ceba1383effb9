#!/bin/bash
# round 6, call 3: the GPU suite from a COLD JIT cache (durations), the cache it leaves, experiments (64-column stage A,
# radix 49, BI batch counts), 8-rank dry runs
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf build/jit_cache
( time python -m pytest tests -m gpu -x -q --durations=40 ) > gpurun_out/r6_durations_cold.txt 2>&1
tail -4 gpurun_out/r6_durations_cold.txt
du -sh build/jit_cache; ls build/jit_cache | wc -l
( time python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r6_durations_warm2.txt 2>&1
tail -4 gpurun_out/r6_durations_warm2.txt
tar czf gpurun_out/r6_jit_cache.tgz -C build jit_cache; ls -la gpurun_out/r6_jit_cache.tgz
O=gpurun_out/r6_exp3.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 600 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_exp3_$tag.err; }
export PERF_STAGE_HX_CASES="f32:68640,f32:62500,f32:120000,f32:250000,f64:68640,f64:250000,f32:30000"
run fpw_default PFFT_JIT_VERBOSE=0
run fpw64 PFFT_JIT_STRIDED_FPW=64
export PERF_STAGE_HX_CASES="f32:16807,f32:2401,f32:117649"
run r49_default PFFT_JIT_VERBOSE=0
run r49 PFFT_JIT_SPEC_RADICES=16807:49x49x7
run r49b PFFT_JIT_SPEC_RADICES=2401:49x49
unset PERF_STAGE_HX_CASES
cat $O
python tools/perf_bi_batches.py > gpurun_out/r6_bi_batches.txt 2>&1
PFFT_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r6_bench_rccl_world1.json 2> gpurun_out/r6_rccl_w1.err
PFFT_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r6_bench_8rank_one_device.json 2> gpurun_out/r6_8rank.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r6_bench_c2_plain.json 2>/dev/null
tail -c 600 gpurun_out/r6_bench_8rank_one_device.json
