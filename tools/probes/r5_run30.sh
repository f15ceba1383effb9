# round 5: pair plans behind any first radix >= 15, the image padded with a period R0 for every even R0 (experiment knobs)
mkdir -p gpurun_out/r5_run30
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run30/perf_hx_pad_even.txt
import os, subprocess, sys
ROOT = os.getcwd()
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
for prec, sizes in (("f32", [10500, 11000, 11250, 12500, 13000, 13500, 14000, 15000, 15625, 22500, 27000, 30000, 33750]),
                    ("f64", [5250, 5500, 5625, 6000, 6250, 6750, 7000, 7500, 10500, 11250, 12000, 13500, 14000])):
    for n in sizes:
        for tag, env in (("default", {}), ("any r0 + even padding", {"PFFT_JIT_HX_PAIR_ANY_R0": "1", "PFFT_JIT_HX_PAD_EVEN": "1", "PFFT_NO_TUNED_TABLE": "1"}),
                         ("any r0", {"PFFT_JIT_HX_PAIR_ANY_R0": "1", "PFFT_NO_TUNED_TABLE": "1"})):
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), tag], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, tag, p.stderr[-300:])), flush=True)
PY
