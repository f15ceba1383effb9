#!/bin/bash
# per-kernel device times of packed 1-D plans (run through gpurun): tools/prof_1d.sh <outdir> <prec:n:batch>...
# rocprofv3 --kernel-trace --stats of tools/probes/one_1d.py; prints the largest kernels with their average time.
# Environment knobs (PFFT_*) pass through, so the same line profiles a forced variant.
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for spec in "$@"; do
  IFS=: read -r prec n batch <<< "$spec"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$spec" -- python3 tools/probes/one_1d.py "$prec" "$n" "$batch" 10 > "$out/$spec.log" 2>&1
  grep "TB/s" "$out/$spec.log" | tail -1
  python3 - "$out/$spec" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:4]:
        print("   %8.1f us avg  x%-5s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:230]))
PY
done
