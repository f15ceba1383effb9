#!/bin/bash
set -u
export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out
python bench.py > $out/r2_bench_c2.json 2> $out/c2.err
for c in c3 c5 ref16 ref256 ref4096 ref65536; do python bench.py --config $c --no-cpu-baseline > $out/r2_bench_$c.json 2> $out/$c.err; done
for c in c2 c5 ref65536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $out/stats_$c.log 2>&1
done
for f in $out/r2_bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], (r.get('copy_probe') or {}).get('gbs'))" $f; done
# PMC traffic passes of the multi-launch configs (kernel names carry the cache-policy AUX value)
for c in c3 c5 ref65536; do tools/run_pmc.sh $c $out/pmc_$c > $out/pmc_$c.log 2>&1; done
