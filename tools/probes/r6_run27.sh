#!/bin/bash
# round 6, call 27: fuzz after the batch-of-one fix and the output-only rule of policy 3 (seeds 97-99, general; 101 stages)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/fuzz.py 97 120 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_97_120.txt; tail -2 gpurun_out/r6_fuzz_97_120.txt
python tools/fuzz.py 98 150 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_98_150.txt; tail -2 gpurun_out/r6_fuzz_98_150.txt
python tools/fuzz.py 99 150 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_99_150.txt; tail -2 gpurun_out/r6_fuzz_99_150.txt
python tools/fuzz.py 101 120 stages 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_101_120_stages.txt; tail -2 gpurun_out/r6_fuzz_101_120_stages.txt
