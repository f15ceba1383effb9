# round 5: transforms per work-group of the two-per-CU plans (1 against 2 / 4 / persistent) through bench.py
mkdir -p gpurun_out/r5_run41
for c in g32_14 g64_13 ref15360; do for g in default 2 4 0; do
  if [ $g = default ]; then unset PFFT_GROUPS_PER_WG; else export PFFT_GROUPS_PER_WG=$g; fi
  python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c gpw=$g', d['ms_per_step'], r['frac'], r.get('frac_wall'), r.get('kernel_ms_min_median_max'))"
done; done 2>&1 | tee gpurun_out/r5_run41/pair_grid_rule.txt
