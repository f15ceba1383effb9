# round 5: LDS-DMA form of the register-resident kernel (tuner), row-lanes stage B, then the whole GPU suite (no -x) and fuzz seed 61 in full
mkdir -p gpurun_out/r5_run22
for c in 32768 16384064 8192064 16387; do timeout 300 build/tune/tune_$c > gpurun_out/r5_run22/tune_$c.txt 2>&1; tail -25 gpurun_out/r5_run22/tune_$c.txt; done
python tools/perf_tin_rows.py 2>&1 | tee gpurun_out/r5_run22/perf_tin_rows.txt
( time timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -60 ) 2>&1 | tee gpurun_out/r5_run22/pytest_all.txt
python tools/fuzz.py 61 150 2>&1 | grep -v "^\.\.\.\|^skip" | tee gpurun_out/r5_run22/fuzz_61_150.txt
tar czf gpurun_out/r5_run22/jit_cache.tgz -C build jit_cache 2>/dev/null
