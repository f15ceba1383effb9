mkdir -p gpurun_out/r4_xval
timeout 900 python -m pytest tests/test_gpu_xcd_local.py tests/test_cpp_facade.py -x -q -m gpu 2>&1 | tail -8
PFFT_XCD_MIN_BATCH=1 PFFT_XCD_CHECK=1 timeout 900 python tools/fuzz.py 21 120 global > gpurun_out/r4_xval/fuzz_global_forced.txt 2>&1; echo "fuzz forced rc $?"; tail -4 gpurun_out/r4_xval/fuzz_global_forced.txt
timeout 900 python tools/fuzz.py 22 60 global > gpurun_out/r4_xval/fuzz_global.txt 2>&1; echo "fuzz rc $?"; tail -3 gpurun_out/r4_xval/fuzz_global.txt
