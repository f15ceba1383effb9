cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide_register_resident or register_resident_stage or strided_workgroup_tier or multidimensional" > gpurun_out/wide_tests.txt 2>&1
tail -15 gpurun_out/wide_tests.txt
{
echo "== fp32 1536 with the 3-register spill (PFFT_JIT_STRIDED_HX_FORCE=64:1) against the two-stage plan"
PFFT_JIT_STRIDED_HX_FORCE=64:1 python tools/perf_bi_wide.py f32 1536
PFFT_NO_BI_WIDE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 1536
echo "== split storage"
PFFT_PERF_SPLIT=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py both 1280 2048
PFFT_PERF_SPLIT=1 PFFT_PERF_QUICK=1 PFFT_NO_BI_WIDE=1 python tools/perf_bi_wide.py both 1280 2048
echo "== other lengths"
PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py both 1152 1200 1250 1296 1344 1440 1500 1600 1792 1875 1936 2000
PFFT_PERF_QUICK=1 PFFT_NO_BI_WIDE=1 python tools/perf_bi_wide.py both 1152 1200 1250 1296 1344 1440 1500 1600 1792 1875 1936 2000
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_second.txt
cat gpurun_out/r6_bi_wide_second.txt
tar czf gpurun_out/jit_cache.tgz -C build jit_cache
