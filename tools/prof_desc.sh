#!/bin/bash
# per-kernel device times of any descriptor (run through gpurun): tools/prof_desc.sh <outdir> <float|double>:<key=value,...> ...
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for spec in "$@"; do
  i=$((i+1))
  prec=${spec%%:*}; desc=${spec#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/d$i" -- python3 tools/probes/one_desc.py "$prec" "$desc" 10 > "$out/d$i.log" 2>&1
  grep " ms" "$out/d$i.log" | tail -1
  python3 - "$out/d$i" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "pfa::" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:4]:
        print("   %8.1f us avg  x%-5s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:230]))
PY
done
