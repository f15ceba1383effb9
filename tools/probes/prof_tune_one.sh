#!/bin/bash
export TMPDIR=/tmp
export TUNE_ONE=1
tools/tune_2d_small
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_tuneone -- tools/tune_2d_small > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/r2_tuneone/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.reader(open(f)))[1:4]:
        print(r[0][:120], "| calls", r[1], "avg ns", r[3], "min", r[5], "max", r[6])
PY
python bench.py --config c5 --steps 30 --no-cpu-baseline | tail -c 400
