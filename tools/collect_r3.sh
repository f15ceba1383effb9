#!/bin/bash
# copy the end-of-round artefacts of tools/final_profiles_r3.sh from gpurun_out/final_r3 into profiles/
set -u
src=gpurun_out/final_r3
cp $src/r3_bench_*.json $src/r3_*_kernel_stats.csv $src/r3_pmc_traffic*.json profiles/ 2>/dev/null
for f in $src/survey/survey_*.txt; do b=$(basename $f); grep -v amdgpu.ids $f > profiles/r3_$b; done
ls profiles/r3_*
