mkdir -p gpurun_out/r5_run9
python tools/perf_hx.py all 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_run9/perf_hx.txt
