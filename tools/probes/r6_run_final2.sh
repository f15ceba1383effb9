cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/suite_final2.txt 2>&1
tail -4 gpurun_out/suite_final2.txt
ONLY_PMC="bi32_2048 bi64_2048" bash tools/final_profiles_r6.sh > gpurun_out/final2_pmc.log 2>&1
tail -5 gpurun_out/final2_pmc.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
tar czf gpurun_out/jit_cache.tgz -C build jit_cache
