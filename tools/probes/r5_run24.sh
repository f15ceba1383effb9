# round 5: register-resident kernel, two work-groups per CU, across the 64 ... 160 KiB band; long strided-layout test
mkdir -p gpurun_out/r5_run24
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "long_transforms_in_strided" 2>&1 | tail -5 ) | tee gpurun_out/r5_run24/pytest_sel.txt
timeout 2400 python tools/perf_hx_pairs.py 2>&1 | tee gpurun_out/r5_run24/perf_hx_pairs.txt
