# the reference's WorkgroupOrGlobal sizes (8192 / 16384, float and double) and their neighbours, 1 GiB
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_wog.txt
import sys; sys.path.insert(0, "tools")
from perf_survey_lib import run
for prec, es in (("f32", 8), ("f64", 16)):
    for n in (4096, 8192, 16384, 32768):
        run("%s N=%d" % (prec, n), [n], (1 << 30) // (n * es), prec)
PY
