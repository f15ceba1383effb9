cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
{
echo "== experiment: 8 fp32 / 4 fp64 columns (64-byte segments) in one pass for N = 2049 ... 4096 (PFFT_BI_WIDE_FPW) against the two-stage plan"
PFFT_BI_WIDE_FPW=8 PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 2560 3072 4096 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
PFFT_BI_WIDE_FPW=4 PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f64 2560 3072 4096 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py both 2560 3072 4096
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_seventh.txt
cat gpurun_out/r6_bi_wide_seventh.txt
