# round 5: the whole GPU suite (no -x) on the final build with the JIT cache of the previous run, fuzz 61 in full
mkdir -p gpurun_out/r5_final build
[ -f gpurun_out_in/jit_cache.tgz ] && tar xzf gpurun_out_in/jit_cache.tgz -C build
( time timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -60 ) 2>&1 | tee gpurun_out/r5_final/pytest_all.txt
python tools/fuzz.py 61 150 2>&1 | grep -v "^\.\.\.\|^skip" | tee gpurun_out/r5_final/fuzz_61_150.txt
tar czf gpurun_out/r5_final/jit_cache.tgz -C build jit_cache 2>/dev/null
