#!/bin/bash
# PMC calibration passes on copy kernels with known byte counts:  tools/run_pmc_cal.sh <binary> <outdir>
set -u
bin=$1; out=$2
mkdir -p "$out"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/$c" -- "$bin" > "$out/$c.log" 2>&1
done
