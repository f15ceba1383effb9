# round 5: where the XCD-local plan with its recovery launch loses against round 4 (variants: PFA_XCD_EXP bits, stockham_xcd.hpp)
mkdir -p gpurun_out/r5_recover
timeout 1500 python -m pytest tests/test_gpu_xcd_local.py -x -q 2>&1 | tail -15 | tee gpurun_out/r5_recover/pytest2.txt
one() { python bench.py --config $1 --no-cpu-baseline --steps ${2:-100} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_ms'], r['launches_per_execute'], d['config']['parity_rel_l2_vs_numpy'])"; }
for rep in 1 2; do
for c in ref65536 g32_17; do
  echo -n "$c new: "; one $c
  echo -n "$c new, no recovery launch: "; PFFT_XCD_NO_RECOVER=1 one $c
  for v in 1 2 4 7; do echo -n "$c exp$v, no recovery launch: "; PFFT_XCD_NO_RECOVER=1 PORTFFT_AMD_LIBRARY=$PWD/build/libpfft_exp$v.so one $c; done
  echo -n "$c r4 : "; PORTFFT_AMD_LIBRARY=$PWD/build/libportfft_amd_r4.so one $c
done; done 2>&1 | tee gpurun_out/r5_recover/ab2.txt
