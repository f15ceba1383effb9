#!/bin/bash
# forced first factors (PFFT_GLOBAL_N1) of four-step sizes: tools/probes/half_pairs_n1.sh "f32:393216 384" ...
for spec in "$@"; do
  set -- $spec
  echo -n "N1=$2 "; PFFT_GLOBAL_N1=$2 python3 tools/probes/half_pairs.py $1 2>&1 | grep "N=" | head -1
done
