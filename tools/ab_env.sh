#!/bin/bash
# A/B of one environment variable over bench.py configs, alternating runs:  tools/ab_env.sh VAR=VALUE "c5 c3" [rounds]
# prints config, setting, ms per execute (event-timed) and fraction of the 8 TB/s peak
set -u
kv=$1; cfgs=$2; rounds=${3:-2}
for i in $(seq 1 "$rounds"); do
  for c in $cfgs; do
    for mode in default "$kv"; do
      if [ "$mode" = default ]; then out=$(python bench.py --config "$c" --steps 30 --no-cpu-baseline 2>/dev/null | tail -1)
      else out=$(env "$kv" python bench.py --config "$c" --steps 30 --no-cpu-baseline 2>/dev/null | tail -1); fi
      python3 -c "
import json,sys
d=json.loads(sys.argv[1]); r=d['roofline']
print('%-9s %-28s kernel_ms %.4f frac %.4f  ms_per_step %.4f' % (sys.argv[2], sys.argv[3], r['kernel_ms'], r['frac'], d['ms_per_step']))" "$out" "$c" "$mode"
    done
  done
done
