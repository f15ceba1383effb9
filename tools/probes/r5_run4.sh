# round 5, run 4: what is left of the XCD launch's loss against round 4 (PFA_XCD_EXP variants); the register-resident tuner
mkdir -p gpurun_out/r5_run4
(./build/tune_32768; ./build/tune_16384064) 2>&1 | tee gpurun_out/r5_run4/tune_hx.txt
one() { python bench.py --config $1 --no-cpu-baseline --steps ${2:-100} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_ms'], r['launches_per_execute'], d['config']['parity_rel_l2_vs_numpy'])"; }
for rep in 1 2; do
for c in g32_17 ref65536 g64_18; do
  echo -n "$c new: "; one $c
  for v in 1 2 4 8 15; do echo -n "$c exp$v: "; PORTFFT_AMD_LIBRARY=$PWD/build/libpfft_exp$v.so one $c; done
  echo -n "$c r4 : "; PORTFFT_AMD_LIBRARY=$PWD/build/libportfft_amd_r4.so one $c
done; done 2>&1 | tee gpurun_out/r5_run4/ab.txt
