export TMPDIR=/tmp
out=gpurun_out/last; mkdir -p $out
for c in g32_21 g32_24; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $out/stats_$c.log 2>&1
  f=$(ls $out/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $out/r3_${c}_kernel_stats.csv
done
ls $out
