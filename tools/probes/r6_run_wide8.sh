cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
{
echo "== experiment: 32 fp32 / 16 fp64 columns (256-byte segments) on ONE register-resident work-group per CU for N <= 1024 (PFFT_BI_WIDE_FPW) against the plans of the default path"
PFFT_BI_WIDE_FPW=32 PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 640 768 800 896 1024 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
PFFT_BI_WIDE_FPW=16 PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f64 640 768 800 896 1024 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
echo "== default"
PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py both 640 768 800 896 1024
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_eighth.txt
cat gpurun_out/r6_bi_wide_eighth.txt
