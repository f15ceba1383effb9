#!/bin/bash
# PMC passes of the XCD-local tuner (tools/tune_xcd.hip) for a few (slots, lag, work-groups per CU) points:
#   tools/probes/xcd_pmc.sh <outdir> <binary>
set -u
out=$1; bin=$2
mkdir -p "$out"
export TUNE_REPS=3
for cfg in "4 8 4" "3 5 4" "3 3 2" "2 2 2" "4 8 3"; do
  set -- $cfg
  export TUNE_SLOTS_LOG2=$1 TUNE_LAG=$2 TUNE_WG_PER_CU=$3
  tag="S$((1<<$1))_lag$2_w$3"
  bash tools/pmc_binary.sh "$out/$tag" "$bin" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" > "$out/$tag.txt" 2>&1
  echo "== $tag"; grep -A1 "xcd_fourstep\|stockham_strided_kernel" "$out/$tag.txt" | cut -c1-300
done
