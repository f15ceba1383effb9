# round 5: the whole GPU suite on the final build (its JIT cache is brought back so that later runs start warm), fp64 8192 profiles
mkdir -p gpurun_out/r5_final
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14 ) 2>&1 | tee gpurun_out/r5_final/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r5_final/smoke.txt
tar czf gpurun_out/r5_final/jit_cache.tgz -C build jit_cache 2>/dev/null; ls -la gpurun_out/r5_final/jit_cache.tgz
ONLY_PMC="g64_13" bash tools/final_profiles_r5.sh > gpurun_out/r5_final/pmc_g64_13.log 2>&1
python tools/fuzz.py 61 150 2>&1 | tail -2 | tee gpurun_out/r5_final/fuzz_61_150.txt
python tools/fuzz.py 62 60 global 2>&1 | tail -2 | tee gpurun_out/r5_final/fuzz_62_60_global.txt
python tools/fuzz.py 63 100 regres 2>&1 | tail -2 | tee gpurun_out/r5_final/fuzz_63_100_regres.txt
