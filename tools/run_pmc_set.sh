#!/bin/bash
# a set of PMC counters (one rocprofv3 pass per group) over one bench.py config, per-kernel means printed:
#   tools/run_pmc_set.sh <config> <outdir> "CTR_A CTR_B" "CTR_C" ...
set -u
cfg=$1; out=$2; shift 2
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/g$i" -- python3 bench.py --config "$cfg" --steps 3 --warmup 1 --no-cpu-baseline > "$out/g$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pfa::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k[:170])
    for c, v in sorted(cs.items()):
        print("   %-44s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
