cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
{
echo "== N = 2000 on the narrowest-widest-pass radices (16.5.5.5: 35 values per lane; the planner's 10.8.5.5 holds 40)"
PFFT_JIT_VERBOSE=1 python tools/perf_bi_wide.py both 2000 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
PFFT_NO_BI_WIDE=1 python tools/perf_bi_wide.py both 2000
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_sixth.txt
cat gpurun_out/r6_bi_wide_sixth.txt
