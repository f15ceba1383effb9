# round 5, run 6: the new tests and the register-resident bench rows
mkdir -p gpurun_out/r5_run6
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan_measure.py tests/test_gpu_xcd_local.py -x -q -s -k "register_resident or every_tuned or maximum_sizes or wave64 or xcd or gives_up" 2>&1 | tail -15 | tee gpurun_out/r5_run6/pytest.txt
one() { python bench.py --config $1 --no-cpu-baseline --steps ${2:-100} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['frac_wall'], r['kernel_ms'], r['launches_per_execute'], d['config']['parity_rel_l2_vs_numpy'])"; }
for rep in 1 2; do
  for c in g32_15 g64_14; do
  echo -n "$c hx : "; one $c
  echo -n "$c two: "; PFFT_NO_REGRES=1 one $c
  done
done 2>&1 | tee gpurun_out/r5_run6/hx.txt
