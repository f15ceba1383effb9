mkdir -p gpurun_out/r4_x21
timeout 600 python -m pytest tests/test_gpu_xcd_local.py -x -q 2>&1 | tail -6
for b in 121_0 122_0; do
  TUNE_SWEEP=1 timeout 300 build/tune_xcd_g_$b > gpurun_out/r4_x21/$b.txt 2>&1
  echo "== $b rc $? bit-identical rows $(grep -c bit-identical gpurun_out/r4_x21/$b.txt) bad $(grep -ci 'mismatch\|timeout' gpurun_out/r4_x21/$b.txt)"
  grep "two launches" gpurun_out/r4_x21/$b.txt | tail -1
  grep "XCD-local" gpurun_out/r4_x21/$b.txt | awk '{for(i=1;i<=NF;i++) if($i=="ms"){print $(i-1), $0}}' | sort -n | head -3 | cut -d' ' -f2-
done
