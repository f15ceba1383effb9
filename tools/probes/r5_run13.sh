mkdir -p gpurun_out/r5_run13
(./build/tune_grid_32768; ./build/tune_grid_16384064) 2>&1 | tee gpurun_out/r5_run13/tune_grid.txt
timeout 900 python -m pytest tests/test_gpu_xcd_local.py tests/test_gpu_parity.py -x -q -k "xcd or gives_up or register_resident or maximum" 2>&1 | tail -3 | tee gpurun_out/r5_run13/pytest.txt
