# round 5: would the two-per-CU plans also pay between 64 and 80 KiB?
mkdir -p gpurun_out/r5_run27
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run27/perf_hx_pairs_64_80.txt
import os, subprocess, sys
ROOT = os.getcwd()
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
for prec, sizes in (("f32", [8448, 8704, 8960, 9216, 9600, 9728, 9984, 10240]), ("f64", [4224, 4352, 4480, 4608, 4800, 4864, 4992, 5120])):
    for n in sizes:
        for tag, env in (("default", {}), ("pair from 64 KiB", {"PFFT_JIT_HX_PAIR_MIN_KIB": "64", "PFFT_NO_TUNED_TABLE": "1", "PFFT_NO_PRECOMPILED": "1"})):
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), tag], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, tag, p.stderr[-300:])), flush=True)
PY
