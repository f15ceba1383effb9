#!/bin/bash
# The survey tables of one round, from one script (run through gpurun): tools/survey.sh <outdir>
# Copy <outdir>/survey_*.txt to profiles/r<N>_survey_*.txt afterwards.
set -u
out=${1:-gpurun_out/survey}; mkdir -p "$out"
for s in global_f32 global_f64 global_np2 global_semi global_long split split_global generic unpacked strided primes; do
  timeout 900 python3 tools/perf_$s.py > "$out/survey_$s.txt" 2> "$out/survey_$s.err" || echo "perf_$s.py failed (rc $?)"
  echo "== $s"; cat "$out/survey_$s.txt"
done
