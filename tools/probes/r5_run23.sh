# round 5: register-resident forms with two work-groups per CU (tuner), long strided-layout transforms, tuned-table tests, fuzz 61
mkdir -p gpurun_out/r5_run23
for c in 16388 8192065 12000; do timeout 300 build/tune/tune_$c > gpurun_out/r5_run23/tune_$c.txt 2>&1; tail -22 gpurun_out/r5_run23/tune_$c.txt; done
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan_measure.py -m gpu -q -k "long_transforms_in_strided or tuned or strided_workgroup or runtime_specialised" 2>&1 | tail -30 ) | tee gpurun_out/r5_run23/pytest_sel.txt
python tools/fuzz.py 61 150 2>&1 | grep -v "^\.\.\.\|^skip" | tee gpurun_out/r5_run23/fuzz_61_150.txt
python tools/fuzz.py 64 150 2>&1 | grep -v "^\.\.\.\|^skip" | tee gpurun_out/r5_run23/fuzz_64_150.txt
