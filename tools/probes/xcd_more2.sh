mkdir -p gpurun_out/r4_x20
timeout 900 python -m pytest tests/test_gpu_xcd_local.py -x -q 2>&1 | tail -15
bash tools/probes/xcd_lib_ab.sh
for b in 119_0 191_0 16_0 116_512 17_0; do
  TUNE_SWEEP=1 timeout 300 build/tune_xcd_g_$b > gpurun_out/r4_x20/$b.txt 2>&1
  echo "== $b rc $? bit-identical rows $(grep -c bit-identical gpurun_out/r4_x20/$b.txt) bad $(grep -ci 'mismatch\|timeout' gpurun_out/r4_x20/$b.txt)"
done
