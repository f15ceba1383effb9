#!/bin/bash
# round 6, last call: does the JIT cache that travels with the tree get used (a prime-factor test that takes 65 s cold)?
# then the end-of-round artefacts
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ls -la build/jit_cache | head -4
( time python -m pytest tests/test_gpu_parity.py -x -q -k "wave64_prime_factors and f32" ) 2>&1 | tail -5 | tee gpurun_out/r6_shipped_cache_check.txt
bash tools/final_profiles_r6.sh > gpurun_out/r6_final_profiles.log 2>&1
tail -30 gpurun_out/r6_final_profiles.log
