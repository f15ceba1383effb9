#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_c11; mkdir -p $O
for c in 20 120 16; do timeout 100 ./build/tune_fourstep_dead_$c > $O/dead_$c.txt 2>&1; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "global_sizes or strided_workgroup" > $O/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $O/pytest1.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cache_sized or config3" > $O/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -3 $O/pytest2.log
for c in c3 c3 ref65536; do python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'])" $O/bench_$c.json; done
PFFT_NO_FS_PAIRS=1 python bench.py --config c3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('c3 without fs pairs', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'])"
