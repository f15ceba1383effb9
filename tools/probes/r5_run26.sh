# round 5: pair plans before registered / tuned LDS-resident kernels, the scratch fallback; parity and the sweep again
mkdir -p gpurun_out/r5_run26
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan_measure.py -m gpu -q -k "register_resident or every_registered or tuned or maximum_sizes or unpacked" 2>&1 | tail -8 ) | tee gpurun_out/r5_run26/pytest_sel.txt
python tools/fuzz.py 72 100 pairs 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_run26/fuzz_72_100_pairs.txt
PFFT_JIT_VERBOSE=1 timeout 1500 python tools/perf_hx_pairs.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run26/perf_hx_pairs.txt
for c in g32_14 g64_13 ref15360 c2; do python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_run26/bench_$c.json; done
