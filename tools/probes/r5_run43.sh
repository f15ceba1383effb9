# round 5: the pre-compiled pair entries through the library (fp32 12288 / 16384, fp64 6144 / 8192) and their LDS-resident twins
mkdir -p gpurun_out/r5_run43
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run43/perf_registered_pairs.txt
import os, subprocess, sys
ROOT = os.getcwd()
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
for prec, sizes in (("f32", [12288, 16384]), ("f64", [6144, 8192])):
    for n in sizes:
        for tag, env in (("lds", {"PFFT_NO_REGRES": "1"}), ("default", {})):
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), tag], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, tag, p.stderr[-300:])), flush=True)
PY
