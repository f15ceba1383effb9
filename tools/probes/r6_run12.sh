#!/bin/bash
# round 6, call 12: 2-D 1000 x 1000 (0.28 against C5's 0.39): per-pass times, forced first-pass shapes, the per-dimension twin;
# BI two-stage + fp32 n=512 adoption check through the suite's strided tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r6_2d_1000.txt; : > $O
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_2d1000_prof -- python3 tools/probes/one_2d.py f32 1000 1000 10 > gpurun_out/r6_2d1000_prof.log 2>&1
python3 - gpurun_out/r6_2d1000_prof <<'PY' >> $O
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:4]:
        print("   %8.1f us avg  x%-5s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:170]))
PY
python tools/jit_sweep_2d.py f32 1000x1000 rows2d 1000:8:100:10x10x10 1000:8:200:10x20x5 1000:8:200:20x10x5 1000:4:100:10x10x10 1000:4:200:10x20x5 1000:4:200:20x10x5 1000:2:100:10x10x10 1000:2:200:10x20x5 1000:8:250:25x10x4 1000:4:250:25x10x4 >> $O 2>&1
python tools/jit_sweep_2d.py f32 1000x1000 strided 125:64:10:5x5x5 125:64:5:25x5 125:64:8:5x5x5 125:32:10:5x5x5 125:64:13:5x5x5 >> $O 2>&1
PFFT_2D_TWO_PASS=0 python tools/perf_2d.py child f32 1000x1000 2>/dev/null | grep TB >> $O
python tools/perf_2d.py child f32 1000x1000 2>/dev/null | grep TB >> $O
cat $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "strided_workgroup_tier or strided_layouts or register_resident_stage" 2>&1 | tail -3
