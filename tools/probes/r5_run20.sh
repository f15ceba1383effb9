mkdir -p gpurun_out/r5_run20
python tools/perf_hx_below.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_run20/perf_hx_below_pf512.txt
