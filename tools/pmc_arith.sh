#!/bin/bash
# SQ counters of the arithmetic-limited kernels: one rocprofv3 --pmc pass per counter group
# (counters alone with --kernel-trace: the guide's recipe), per-kernel means + the derived figures the question needs:
#   VALU issue utilisation = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES-equivalent, waves per SIMD, wait shares, LDS conflicts,
#   registers / scratch / LDS from the kernel trace.
#   tools/pmc_arith.sh <outdir>                                  the round-4 set
#   tools/pmc_arith.sh <outdir> tag n batch prec [tag n batch prec ...]   other descriptors
set -u
out=$1; mkdir -p "$out"
export TMPDIR=/tmp PFFT_JIT_CACHE_DIR=/tmp/pmc_arith_cache
CTR_GROUPS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"
        "GRBM_GUI_ACTIVE")
run_case() {  # tag n batch prec   (prec "f32bi" / "f64bi": batch-interleaved on both sides)
  local tag=$1 n=$2 b=$3 prec=${4%bi} i=0 lay=""
  [ "$4" != "$prec" ] && lay=bi
  for grp in "${CTR_GROUPS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/$tag/g$i" -- python3 tools/probes/one_desc.py $n $b $prec 3 $lay > "$out/$tag.g$i.log" 2>&1
  done
  python3 tools/summarize_sq.py "$out/$tag" "$tag" > "$out/$tag.txt" 2>&1
  cat "$out/$tag.txt"
  rm -rf "$out/$tag"
}
if [ $# -ge 5 ]; then  # cases on the command line: <outdir> tag n batch prec [tag n batch prec ...]
  shift
  while [ $# -ge 4 ]; do run_case "$1" "$2" "$3" "$4"; shift 4; done
  exit 0
fi
run_case f32_1e6 1000000 128 f32
run_case f32_62500 62500 2048 f32
run_case f32_30000 30000 4096 f32
run_case f32_43x47 2021 66000 f32
run_case f32_61x16 976 137000 f32
run_case f32_2e20 1048576 128 f32   # (the power-of-two four-step pair, for comparison)
run_case f32_4096 4096 32768 f32    # (the headline kernel, for comparison)
