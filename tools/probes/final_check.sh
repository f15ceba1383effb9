# round 5: the whole GPU suite and the default bench on the last build (jit.cpp split in two sources, jit_knobs)
mkdir -p gpurun_out/r5_final6
( time timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r5_final6/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r5_final6/smoke.txt
python tools/fuzz.py 70 100 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_final6/fuzz_70_100.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r5_final6/bench_default.json
