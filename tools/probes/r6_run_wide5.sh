cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
{
echo "== 36 fp32 values per lane (PFFT_JIT_STRIDED_HX_WIDE_SLACK=8, the default): lengths the register estimate had excluded"
PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 1728 1800 1920 1944 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f64 1920 2000 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
echo "== two-stage twins"
PFFT_NO_BI_WIDE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 1728 1800 1920 1944
PFFT_NO_BI_WIDE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f64 1920 2000
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_fifth.txt
cat gpurun_out/r6_bi_wide_fifth.txt
