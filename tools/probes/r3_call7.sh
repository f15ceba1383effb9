#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_c7; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "global_sizes or strided_workgroup or maximum or reference_size_grid" > $O/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $O/pytest1.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cache_sized or config3 or beyond_4gib" > $O/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -3 $O/pytest2.log
python3 tools/perf_global_f32.py > $O/survey_global_f32.txt 2>&1; cat $O/survey_global_f32.txt
PFFT_NO_FS_PAIRS=1 python3 tools/perf_global_f32.py > $O/survey_global_f32_nofs.txt 2>&1; cat $O/survey_global_f32_nofs.txt
for c in ref65536 c3; do python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'])" $O/bench_$c.json; done
