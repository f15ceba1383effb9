# round 5: the register-resident test with the second-choice pair plans added
mkdir -p gpurun_out/r5_run42
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "register_resident" 2>&1 | tail -5 ) | tee gpurun_out/r5_run42/pytest_sel.txt
