cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
L="1100 1125 1176 1225 1323 1331 1375 1400 1458 1568 1575 1600 1617 1625 1664 1701 1715 1750 1760 1815 1875"
{
echo "== fp32, scratch tolerated up to 128 B per lane (PFFT_JIT_STRIDED_HX_WIDE_SCRATCH=128); verbose lines name the scratch of the rejected ones"
PFFT_JIT_STRIDED_HX_WIDE_SCRATCH=128 PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 $L 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
echo "== the default tolerance"
PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 $L 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
echo "== two-stage twins"
PFFT_NO_BI_WIDE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 $L
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_fourth.txt
cat gpurun_out/r6_bi_wide_fourth.txt
