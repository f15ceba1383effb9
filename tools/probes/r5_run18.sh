mkdir -p gpurun_out/r5_run18
(./build/tune_8192064; ./build/tune_16387) 2>&1 | tee gpurun_out/r5_run18/tune_pf.txt
