# round 5: the whole GPU suite on the build with the two-per-CU plans, smoke, fuzz, profiles of the changed configs
mkdir -p gpurun_out/r5_final3
( time timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 ) 2>&1 | tee gpurun_out/r5_final3/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r5_final3/smoke.txt
python tools/fuzz.py 65 150 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_final3/fuzz_65_150.txt
python tools/fuzz.py 66 60 global 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_final3/fuzz_66_60_global.txt
ONLY_PMC="g32_14 g64_13 ref15360 g32_15 g64_14" bash tools/final_profiles_r5.sh > gpurun_out/r5_final3/pmc.log 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/r5_final3/bench_default.json
