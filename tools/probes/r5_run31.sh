# round 5: pair plans behind any first radix >= 15 (even ones padded): sweep against the LDS-resident plans, parity
mkdir -p gpurun_out/r5_run31
python tools/fuzz.py 73 120 pairs 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_run31/fuzz_73_120_pairs.txt
timeout 1500 python tools/perf_hx_pairs.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run31/perf_hx_pairs_any_r0.txt
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan_measure.py -m gpu -q -k "register_resident or tuned or maximum_sizes or runtime_specialised" 2>&1 | tail -8 ) | tee gpurun_out/r5_run31/pytest_sel.txt
