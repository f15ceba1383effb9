#!/bin/bash
# round 6, call 8: fuzz on this round's build -- the stage band, the general mode, the four-step sizes, the two-per-CU band
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/fuzz.py 91 150 stages 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_91_150_stages.txt; tail -2 gpurun_out/r6_fuzz_91_150_stages.txt
python tools/fuzz.py 92 120 stages 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_92_120_stages.txt; tail -2 gpurun_out/r6_fuzz_92_120_stages.txt
python tools/fuzz.py 93 200 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_93_200.txt; tail -2 gpurun_out/r6_fuzz_93_200.txt
python tools/fuzz.py 94 80 global 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_94_80_global.txt; tail -2 gpurun_out/r6_fuzz_94_80_global.txt
python tools/fuzz.py 95 80 pairs 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_95_80_pairs.txt; tail -2 gpurun_out/r6_fuzz_95_80_pairs.txt
python tools/fuzz.py 96 60 big2d 2>&1 | grep -v "^\.\.\.\|amdgpu" > gpurun_out/r6_fuzz_96_60_big2d.txt; tail -2 gpurun_out/r6_fuzz_96_60_big2d.txt
