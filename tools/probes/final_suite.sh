#!/bin/bash
# the GPU suite on the tree's JIT cache, with the cache packed for the way back (run through gpurun)
cd "$GRAFT_REPO_ROOT" || exit 1
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/suite_last.txt 2>&1
tail -3 gpurun_out/suite_last.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench c2 frac', d['roofline']['frac'], 'value', d['value'])"
tar czf gpurun_out/jit_cache.tgz -C build jit_cache
