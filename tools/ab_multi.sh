#!/bin/bash
# several env settings over bench configs: tools/ab_multi.sh "cfgs" "ENV1=a ENV2=b" "ENV3=c" ...
cfgs=$1; shift
for c in $cfgs; do
  for kv in default "$@"; do
    if [ "$kv" = default ]; then out=$(python bench.py --config "$c" --steps 30 --no-cpu-baseline 2>/dev/null | tail -1)
    else out=$(env $kv python bench.py --config "$c" --steps 30 --no-cpu-baseline 2>/dev/null | tail -1); fi
    python3 -c "
import json,sys
d=json.loads(sys.argv[1]); r=d['roofline']
print('%-9s %-50s kernel_ms %.4f frac %.4f  ms_per_step %.4f' % (sys.argv[2], sys.argv[3], r['kernel_ms'], r['frac'], d['ms_per_step']))" "$out" "$c" "$kv"
  done
done
