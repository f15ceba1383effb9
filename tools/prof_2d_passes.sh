#!/bin/bash
# per-pass device times of the 2-D plans of a few shapes (run through gpurun): tools/prof_2d_passes.sh <outdir> <prec:shape>...
# rocprofv3 --kernel-trace --stats of tools/perf_2d.py's child mode, one shape per run; prints the two largest kernels.
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for ps in "$@"; do
  prec=${ps%%:*}; shape=${ps##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$ps" -- python3 tools/perf_2d.py child "$prec" "$shape" > "$out/$ps.log" 2>&1
  grep "TB/s" "$out/$ps.log" | tail -1
  python3 - "$out/$ps" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:3]:
        print("   %8.1f us avg  x%-5s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:150]))
PY
done
