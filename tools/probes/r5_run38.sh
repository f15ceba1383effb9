# round 5: runtime-specialised four-step stages with narrower groups so that TWO work-groups share a CU (PFFT_JIT_STRIDED_LDS_KIB)
mkdir -p gpurun_out/r5_run38
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run38/perf_stage_groups.txt
import os, subprocess, sys
ROOT = os.getcwd()
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
for prec, sizes in (("f32", [1000000, 68640, 500000, 2985984, 250000, 100000]), ("f64", [1000000, 68640, 250000])):
    for n in sizes:
        for tag, env in (("default", {}), ("groups <= 80 KiB", {"PFFT_JIT_STRIDED_LDS_KIB": "80"}), ("groups <= 52 KiB", {"PFFT_JIT_STRIDED_LDS_KIB": "52"})):
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), tag], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, tag, p.stderr[-300:])), flush=True)
PY
