#!/bin/bash
# round 6, call 10: the split-storage chunked 2-D plan (new test + C5 split timing), XCD entries on a second box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "split_storage_in_cache or two_pass_2d_plan or cache_sized" 2>&1 | tail -5
for rep in 1 2 3; do
python3 tools/probes/one_2d_split.py f32 1024 1024 split 20 2>/dev/null | grep -v amdgpu
PFFT_NO_SPLIT_2D_CACHED=1 python3 tools/probes/one_2d_split.py f32 1024 1024 split 20 2>/dev/null | grep -v amdgpu | sed 's/$/  (streamed twin)/'
python3 tools/probes/one_2d_split.py f32 1024 1024 interleaved 20 2>/dev/null | grep -v amdgpu
done
python3 tools/probes/one_2d_split.py f64 1024 1024 split 20 2>/dev/null | grep -v amdgpu
PFFT_NO_SPLIT_2D_CACHED=1 python3 tools/probes/one_2d_split.py f64 1024 1024 split 20 2>/dev/null | grep -v amdgpu | sed 's/$/  (streamed twin)/'
python3 tools/probes/one_2d_split.py f32 2048 2048 split 20 2>/dev/null | grep -v amdgpu
PFFT_NO_SPLIT_2D_CACHED=1 python3 tools/probes/one_2d_split.py f32 2048 2048 split 20 2>/dev/null | grep -v amdgpu | sed 's/$/  (streamed twin)/'
bash tools/probes/xcd_entries_ab.sh gpurun_out/r6_xcd_entries_ab_box2.txt
