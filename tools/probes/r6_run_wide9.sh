cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
L="528 540 576 600 625 640"
{
echo "== groups of 64 ... 80 KiB (N = 513 ... 640 at 16 fp32 / 8 fp64 columns): default (LDS-resident, one work-group per CU)"
PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py both $L
echo "== 32 / 16 columns on one register-resident work-group per CU (PFFT_BI_WIDE_FPW)"
PFFT_BI_WIDE_FPW=32 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f32 $L
PFFT_BI_WIDE_FPW=16 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py f64 $L
echo "== two register-resident work-groups per CU from 60 KiB (PFFT_JIT_STRIDED_HX_MIN_KIB=60)"
PFFT_JIT_STRIDED_HX_MIN_KIB=60 PFFT_JIT_VERBOSE=1 PFFT_PERF_QUICK=1 python tools/perf_bi_wide.py both $L 2>&1 | grep -v "^\[portfft_amd jit\] [0-9.]* s,"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_ninth.txt
cat gpurun_out/r6_bi_wide_ninth.txt
