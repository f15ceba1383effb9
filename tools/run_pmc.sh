#!/bin/bash
# rocprofv3 passes of one bench.py config on the GPU box (run through gpurun):  tools/run_pmc.sh <config> <outdir>
#   <outdir>/FETCH_SIZE/runc, <outdir>/WRITE_SIZE/runc  -- one --pmc pass per counter (counters alone, with
#                                                         --kernel-trace only: the guide's recipe)
#   <outdir>/stats/runc                                 -- --kernel-trace --stats (per-kernel durations)
# The program after `--` is python3 itself (no shell / env hop under the profiler).
set -u
cfg=$1; out=$2
mkdir -p "$out"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/$c" -- python3 bench.py --config "$cfg" --steps 5 --warmup 2 --no-cpu-baseline > "$out/$c.log" 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --config "$cfg" --steps 20 --warmup 3 --no-cpu-baseline > "$out/stats.log" 2>&1
tail -c 300 "$out/stats.log"
