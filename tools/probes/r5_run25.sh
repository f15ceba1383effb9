# round 5: two register-resident work-groups per CU -- the planner's rule against the LDS-resident plans, the registered fp32 16384
# entry, parity (fuzz `pairs`, the register-resident tests, every registered length), bench g32_14
mkdir -p gpurun_out/r5_run25
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "register_resident or every_registered or golden or maximum_sizes or row_lanes or unpacked or runtime_specialised" 2>&1 | tail -8 ) | tee gpurun_out/r5_run25/pytest_sel.txt
python tools/fuzz.py 71 120 pairs 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_run25/fuzz_71_120_pairs.txt
timeout 1500 python tools/perf_hx_pairs.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run25/perf_hx_pairs.txt
python bench.py --config g32_14 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/r5_run25/bench_g32_14.json
python bench.py --config ref15360 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/r5_run25/bench_ref15360.json
