#!/bin/bash
# alternating A/B of one environment variable over one bench.py config: bench_ab.sh <config> <VAR=value> [rounds]
cfg=$1; var=$2; rounds=${3:-3}
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['ms_per_step'], r['kernel_ms'], r['frac'])" "$1"; }
for i in $(seq $rounds); do
  python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | show "$cfg default    "
  env $var python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | show "$cfg $var"
done
