# generalised XCD-local kernel, further sizes (tools/tune_xcd.hip cases), slots/lag/WG-per-CU sweep
mkdir -p gpurun_out/r4_x19
for b in 116_0 116_512 118_0 118_1024 120_0 19_0 20_0 15_0 15_512; do
  TUNE_SWEEP=1 timeout 240 build/tune_xcd_g_$b > gpurun_out/r4_x19/$b.txt 2>&1
  echo "== $b rc $?"; grep -E "^N =|two launches" gpurun_out/r4_x19/$b.txt | tail -2
  grep "XCD-local" gpurun_out/r4_x19/$b.txt | sort -t= -k5 -n | awk '{print}' | sort -k13 -n | head -3
  grep -c "bit-identical" gpurun_out/r4_x19/$b.txt; grep -ci "mismatch\|timeout" gpurun_out/r4_x19/$b.txt
done
