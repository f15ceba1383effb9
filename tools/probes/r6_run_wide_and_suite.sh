cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
(python tools/perf_bi_wide.py both 1280 1536 2048; PFFT_NO_BI_WIDE=1 python tools/perf_bi_wide.py both 1280 1536 2048) > gpurun_out/r6_bi_wide_first.txt 2>&1
tail -40 gpurun_out/r6_bi_wide_first.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/suite_cold.txt 2>&1
tail -5 gpurun_out/suite_cold.txt
tar czf gpurun_out/jit_cache.tgz -C build jit_cache; ls -la gpurun_out/jit_cache.tgz
