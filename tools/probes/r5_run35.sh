# round 5: the new pre-compiled pair entries (parity), SQ counters of pair against LDS-resident (fp32 16384, 15360), bench labels
mkdir -p gpurun_out/r5_run35
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "every_registered or register_resident or unpacked" 2>&1 | tail -4 ) | tee gpurun_out/r5_run35/pytest_sel.txt
tools/pmc_arith.sh gpurun_out/r5_run35/sq pair_f32_16384 16384 8192 f32 pair_f32_15360 15360 8704 f32 pair_f64_8192 8192 8192 f64 > gpurun_out/r5_run35/sq_pairs.txt 2>&1
PFFT_JIT_HX_PAIRS=0 PFFT_NO_REGRES=1 tools/pmc_arith.sh gpurun_out/r5_run35/sq_lds lds_f32_16384 16384 8192 f32 lds_f32_15360 15360 8704 f32 lds_f64_8192 8192 8192 f64 > gpurun_out/r5_run35/sq_lds.txt 2>&1
for c in g64_13 g32_14 ref15360; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_run35/r5_bench_$c.json; done
python tools/commit_latency.py 2>&1 | tail -30 > gpurun_out/r5_run35/commit_latency.txt
