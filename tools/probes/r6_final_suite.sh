#!/bin/bash
# round 6: the whole GPU suite on the final build from an EMPTY JIT cache, again on the cache that run left (which is then
# packed for build/jit_cache of the tree), smoke(), the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf build/jit_cache
( time python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r6_final_suite_cold.txt 2>&1
tail -4 gpurun_out/r6_final_suite_cold.txt
( time python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r6_final_suite_warm.txt 2>&1
tail -4 gpurun_out/r6_final_suite_warm.txt
du -sh build/jit_cache; ls build/jit_cache | wc -l
tar czf gpurun_out/r6_jit_cache.tgz -C build jit_cache; ls -la gpurun_out/r6_jit_cache.tgz
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; tail -c 400 gpurun_out/r6_bench_default.json
