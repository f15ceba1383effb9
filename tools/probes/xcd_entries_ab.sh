#!/bin/bash
# The seven registered XCD-local entries against their two-launch twins on the current build (VERDICT r5 task 7): 7 alternating
# runs each of bench.py (100 steps, event-timed fraction of 8 TB/s), medians.  fp32 2^20 needs >= 192 transforms: --manual.
cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/xcd_entries_ab.txt}
frac() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['roofline']['frac'])"; }
one() {  # label, bench args...
  label=$1; shift
  a=(); b=()
  for rep in 1 2 3 4 5 6 7; do
    a+=($(python bench.py "$@" --no-cpu-baseline --steps 100 2>/dev/null | frac))
    b+=($(PFFT_NO_XCD_LOCAL=1 python bench.py "$@" --no-cpu-baseline --steps 100 2>/dev/null | frac))
  done
  python - "$label" "${a[*]}" "${b[*]}" <<'PY'
import sys, statistics as st
x = [float(v) for v in sys.argv[2].split()]; y = [float(v) for v in sys.argv[3].split()]
mx, my = st.median(x), st.median(y)
print("%-10s single launch median %.4f (min %.4f max %.4f)  two launches median %.4f (min %.4f max %.4f)  %+.1f %%" % (sys.argv[1], mx, min(x), max(x), my, min(y), max(y), 100 * (mx / my - 1)), flush=True)
PY
}
{
one ref65536 --config ref65536
one g32_17 --config g32_17
one g32_18 --config g32_18
one g32_20x256 --manual d=cpx,n=1048576,b=256 --precision float
one g64_16 --config g64_16
one g64_17 --config g64_17
one g64_18 --config g64_18
} 2>&1 | tee $out
