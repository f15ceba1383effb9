#!/bin/bash
# experiment: runtime-specialised stage A for REGISTERED lengths without a stage-A entry of stage B's width
# (PFFT_HALF_PAIR_ANY=1 + PFFT_GLOBAL_N1): tools/probes/hp_any.sh "f64:65536 128" ...
for spec in "$@"; do
  set -- $spec
  echo -n "any N1=$2 "; PFFT_HALF_PAIR_ANY=1 PFFT_GLOBAL_N1=$2 python3 tools/probes/half_pairs.py $1 | head -1
done
