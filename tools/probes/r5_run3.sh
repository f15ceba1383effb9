# round 5, run 3: wave-ends-on-give-up form of the XCD launch against round 4; first numbers of the register-resident 2^15 kernel
mkdir -p gpurun_out/r5_run3
timeout 900 python -m pytest tests/test_gpu_xcd_local.py -x -q 2>&1 | tail -5 | tee gpurun_out/r5_run3/pytest_xcd.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "global_sizes or backward_grid" 2>&1 | tail -15 | tee gpurun_out/r5_run3/pytest_global.txt
one() { python bench.py --config $1 --no-cpu-baseline --steps ${2:-100} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_ms'], r['launches_per_execute'], d['config']['parity_rel_l2_vs_numpy'])"; }
for rep in 1 2; do
  echo -n "g32_15 hx : "; one g32_15
  echo -n "g32_15 two: "; PFFT_NO_REGRES=1 one g32_15
done 2>&1 | tee gpurun_out/r5_run3/hx.txt
for rep in 1 2; do
for c in ref65536 g32_17 g32_18 g64_16 g64_18; do
  echo -n "$c new: "; one $c
  echo -n "$c r4 : "; PORTFFT_AMD_LIBRARY=$PWD/build/libportfft_amd_r4.so one $c
done; done 2>&1 | tee gpurun_out/r5_run3/ab.txt
