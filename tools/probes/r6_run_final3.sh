cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/suite_final3.txt 2>&1
tail -3 gpurun_out/suite_final3.txt
{
echo "== unaligned batch counts on the wide groups: policy 3 (default cache policies, the default) against the streamed kernels (PFFT_NO_UNALIGNED_POLICY=1)"
python tools/perf_bi_wide.py both 1280 2048 | grep -v "batch 33000"
PFFT_NO_UNALIGNED_POLICY=1 python tools/perf_bi_wide.py both 1280 2048 | grep -v "batch 33000"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_eleventh.txt
cat gpurun_out/r6_bi_wide_eleventh.txt
python bench.py --no-cpu-baseline | tail -c 700
tar czf gpurun_out/jit_cache.tgz -C build jit_cache
