#!/bin/bash
# round 6, call 4: register-resident plans against registered one-per-CU strided entries (BI 1024 / 2048), C5 in split storage:
# copy yardstick + per-pass kernel times
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_exp4.txt; : > $O
run() { tag=$1; shift; env "$@" timeout 600 python tools/perf_stage_hx.py "$tag" >> $O 2>gpurun_out/r6_exp4_$tag.err; }
export PERF_STAGE_HX_CASES="f32:bi1024,f32:bi1024@132000,f32:bi1024@400000,f64:bi1024,f64:bi1024@66000,f32:bi2048@66000,f64:bi2048@33000,f32:bi768@176000,f32:bi660@200000"
run registered PFFT_JIT_VERBOSE=0
run hx_over_reg PFFT_HX_OVER_REGISTERED=1
run registered2 PFFT_JIT_VERBOSE=0
run hx_over_reg2 PFFT_HX_OVER_REGISTERED=1
unset PERF_STAGE_HX_CASES
cat $O
hipcc -O3 --offload-arch=gfx950 tools/probes/copy_c5_split.hip -o build/copy_c5_split 2>/dev/null
./build/copy_c5_split > gpurun_out/r6_copy_c5_split.txt 2>&1; cat gpurun_out/r6_copy_c5_split.txt
export TMPDIR=/tmp
for st in interleaved split; do
  python3 tools/probes/one_2d_split.py f32 1024 1024 $st 20 >> gpurun_out/r6_c5_split_passes.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_c5_$st -- python3 tools/probes/one_2d_split.py f32 1024 1024 $st 20 > gpurun_out/r6_c5_$st.log 2>&1
  python3 - gpurun_out/r6_c5_$st <<'PY' >> gpurun_out/r6_c5_split_passes.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:3]:
        print("   %8.1f us avg  x%-5s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:170]))
PY
done
cat gpurun_out/r6_c5_split_passes.txt
