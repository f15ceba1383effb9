# round 5: the XCD-local plan with its recovery launch (this tree) against the round-4 library (build/libportfft_amd_r4.so)
mkdir -p gpurun_out/r5_recover
timeout 1500 python -m pytest tests/test_gpu_xcd_local.py -x -q 2>&1 | tail -15 | tee gpurun_out/r5_recover/pytest.txt
one() { python bench.py --config $1 --no-cpu-baseline --steps ${2:-100} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_ms'], r['launches_per_execute'], d['config']['parity_rel_l2_vs_numpy'])"; }
for rep in 1 2 3; do
for c in ref65536 g32_17 g32_18 g64_16 g64_18; do
  echo -n "$c new: "; one $c
  echo -n "$c r4 : "; PORTFFT_AMD_LIBRARY=$PWD/build/libportfft_amd_r4.so one $c
done; done 2>&1 | tee gpurun_out/r5_recover/ab.txt
