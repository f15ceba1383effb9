#!/bin/bash
# round 6, call 1: durations of the GPU suite on the round-5 build (warm JIT cache), baselines of the stage-kernel sizes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=60 ) > gpurun_out/r6_durations_warm.txt 2>&1
python tools/perf_global_np2.py > gpurun_out/r6_np2_base.txt 2>&1
for c in ref68640 g32_24; do python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline >> gpurun_out/r6_bench_base.txt 2>&1; done
tail -5 gpurun_out/r6_durations_warm.txt; cat gpurun_out/r6_np2_base.txt
