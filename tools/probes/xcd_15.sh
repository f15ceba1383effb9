mkdir -p gpurun_out/r4_x22
for b in 15_0 15_512 16_0; do
  TUNE_SWEEP=1 timeout 400 build/tune_xcd_g_$b > gpurun_out/r4_x22/$b.txt 2>&1
  echo "== $b rc $? bit-identical rows $(grep -c bit-identical gpurun_out/r4_x22/$b.txt) bad $(grep -ci 'mismatch\|timeout' gpurun_out/r4_x22/$b.txt)"
  grep "two launches" gpurun_out/r4_x22/$b.txt | tail -1
  grep "XCD-local" gpurun_out/r4_x22/$b.txt | awk '{for(i=1;i<=NF;i++) if($i=="ms"){print $(i-1), $0}}' | sort -n | head -4 | cut -d' ' -f2-
done
