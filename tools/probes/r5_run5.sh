mkdir -p gpurun_out/r5_run5
(./build/tune_16384064) 2>&1 | tee gpurun_out/r5_run5/tune_hx64.txt
