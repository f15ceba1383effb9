#!/bin/bash
# copy the end-of-round artefacts of tools/final_profiles_r4.sh from gpurun_out/final_r4 into profiles/
set -u
src=gpurun_out/final_r4
cp $src/r4_bench_*.json $src/r4_*_kernel_stats.csv $src/r4_pmc_traffic*.json $src/r4_commit_latency.txt profiles/ 2>/dev/null
for f in $src/survey/survey_*.txt; do b=$(basename $f); grep -v amdgpu.ids $f > profiles/r4_$b; done
ls profiles/r4_*
