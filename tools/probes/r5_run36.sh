# round 5: padding of the fp64 pair kernel's 128-bit exchange image (tuner), then the whole suite on the last commit
mkdir -p gpurun_out/r5_run36
timeout 300 build/tune/tune_8192066 2>&1 | tee gpurun_out/r5_run36/tune_8192066.txt | tail -20
( time timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r5_run36/pytest_all.txt
