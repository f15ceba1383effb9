# library A/B: the XCD-local plan against the two-launch plan of the same descriptor, bench.py configs
mkdir -p gpurun_out/r4_xlib
one() { python bench.py --config $1 --no-cpu-baseline --steps ${2:-100} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_ms'], r['launches_per_execute'], d['config']['parity_rel_l2_vs_numpy'])"; }
for c in ref65536 g32_17 g32_18 g32_19 g32_20 g64_16 g64_17 g64_18 c3; do
  echo -n "$c xcd: "; PFFT_XCD_CHECK=0 one $c
  echo -n "$c two: "; PFFT_NO_XCD_LOCAL=1 one $c
done 2>&1 | tee gpurun_out/r4_xlib/ab.txt
