# round 5: runtime-specialised four-step lengths whose stage B cannot read tiles -- row-staged input beyond 512 points?
mkdir -p gpurun_out/r5_rowin
one() { python - "$@" <<'PY'
import os, sys
sys.path.insert(0, "tools")
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%s N=%d %s" % (prec, n, os.environ.get("TAG", "")), [n], max(1, (1 << 30) // (n * es)), prec)
PY
}
for n in 1000000 62500 68640 100000 250000 500000 2985984; do
  TAG=default one f32 $n
  TAG=row1024 PFFT_ROW_IN_MAX_N=1100 one f32 $n
  TAG=row2048 PFFT_ROW_IN_MAX_N=2100 one f32 $n
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_rowin/rowin.txt
