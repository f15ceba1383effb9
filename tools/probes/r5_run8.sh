# round 5, run 8: the whole GPU suite on the split plan_*.cpp + knobs build; C++ tests
mkdir -p gpurun_out/r5_run8
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -20 ) 2>&1 | tee gpurun_out/r5_run8/pytest_all.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/r5_run8/bench_c2.json | cut -c1-400
