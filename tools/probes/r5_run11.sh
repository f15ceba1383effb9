mkdir -p gpurun_out/r5_run11
python tools/perf_hx_below.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_run11/perf_hx_below.txt
python tools/fuzz.py 51 120 regres 2>&1 | tail -4 | tee gpurun_out/r5_run11/fuzz_regres.txt
