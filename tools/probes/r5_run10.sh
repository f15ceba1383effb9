mkdir -p gpurun_out/r5_run10
python tools/perf_hx.py all 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_run10/perf_hx.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "register_resident" 2>&1 | tail -3 | tee gpurun_out/r5_run10/pytest.txt
