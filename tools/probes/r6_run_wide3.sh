cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
{
echo "== fp32 lengths whose wide kernel needs 8 ... 44 B of scratch per lane (tolerated: PFFT_JIT_STRIDED_HX_WIDE_SCRATCH, default 48) against the two-stage plan"
PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 1500 1536 1600 1792 1875 1936
echo "== bench.py --config bi32_2048 / bi64_2048"
python bench.py --config bi32_2048 --steps 20 --no-cpu-baseline | tail -1
python bench.py --config bi64_2048 --steps 20 --no-cpu-baseline | tail -1
PFFT_NO_BI_WIDE=1 python bench.py --config bi32_2048 --steps 20 --no-cpu-baseline | tail -1
PFFT_NO_BI_WIDE=1 python bench.py --config bi64_2048 --steps 20 --no-cpu-baseline | tail -1
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_third.txt
cat gpurun_out/r6_bi_wide_third.txt
tar czf gpurun_out/jit_cache.tgz -C build jit_cache
