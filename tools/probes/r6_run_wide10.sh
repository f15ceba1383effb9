cd $GRAFT_REPO_ROOT
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
{
echo "== the column rule (groups of 74 ... 80 KiB register-resident, two per CU) on the default path; twin PFFT_JIT_STRIDED_HX_COLUMN_RULE=0"
python tools/perf_bi_wide.py both 600 625 640
PFFT_JIT_STRIDED_HX_COLUMN_RULE=0 python tools/perf_bi_wide.py both 600 625 640
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_bi_wide_tenth.txt
cat gpurun_out/r6_bi_wide_tenth.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide_register_resident or register_resident_stage or strided_workgroup_tier or multidimensional or reference_size_grid or random_descriptors or strided_layouts" 2>&1 | tail -3
timeout 800 python tools/fuzz.py 201 150 wide > gpurun_out/r6_fuzz_201_150_wide.txt 2>&1; tail -4 gpurun_out/r6_fuzz_201_150_wide.txt
tar czf gpurun_out/jit_cache.tgz -C build jit_cache
