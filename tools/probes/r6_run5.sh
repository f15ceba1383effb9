#!/bin/bash
# round 6, call 5: four-step splits re-measured on this round's stage kernels (PFFT_PLAN_MEASURE: every candidate split, five
# best per length), C5 split yardstick (fixed pass 2), the whole tuned table through the GPU test
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
./build/copy_c5_split > gpurun_out/r6_copy_c5_split.txt 2>&1; cat gpurun_out/r6_copy_c5_split.txt
TUNE_SHOW=5 TUNE_GLOBAL=62500,68640,100000,120000,250000,500000,1000000,2985984,60000,84000,200000 python tools/gen_tuned_table.py gpurun_out/r6_tuned_splits.inc > gpurun_out/r6_tuned_splits.txt 2>&1
cat gpurun_out/r6_tuned_splits.txt | tail -150
( time PFFT_TEST_ALL_TUNED=1 python -m pytest tests/test_gpu_plan_measure.py -x -q -k every_tuned 2>&1 | tail -3 ) > gpurun_out/r6_tuned_table_all_entries.txt 2>&1
cat gpurun_out/r6_tuned_table_all_entries.txt
