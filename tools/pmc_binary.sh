#!/bin/bash
# PMC counters of a stand-alone tuner binary, one rocprofv3 pass per counter group, per-kernel means printed:
#   tools/pmc_binary.sh <outdir> <binary> "CTR_A CTR_B" "CTR_C" ...
set -u
out=$1; bin=$2; shift 2
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/g$i" -- "$bin" > "$out/g$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pfa::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for cs in agg.values() for c in cs})
print("counters:", " ".join(names))
for k, cs in sorted(agg.items()):
    print(k[:200])
    print("   " + "  ".join("%s=%.4g" % (c, sum(cs[c]) / len(cs[c])) for c in names if c in cs))
PY
