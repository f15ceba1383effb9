#!/bin/bash
# A/B of environment settings over 2-D shapes (tools/perf_2d.py child): tools/ab_2d_env.sh "f32:1200x1200 f64:1000x1000" "VAR=a" "VAR=b VAR2=c" ...
shapes=$1; shift
for ps in $shapes; do
  prec=${ps%%:*}; shape=${ps##*:}
  for kv in default "$@"; do
    if [ "$kv" = default ]; then out=$(python tools/perf_2d.py child "$prec" "$shape" 2>/dev/null | grep "TB/s" | tail -1)
    else out=$(env $kv python tools/perf_2d.py child "$prec" "$shape" 2>/dev/null | grep "TB/s" | tail -1); fi
    echo "$kv | $out"
  done
done
