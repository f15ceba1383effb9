"""Would the register-resident kernel also beat the LDS-resident one for long transforms that still fit the LDS?
PFFT_JIT_HX_MIN_KIB lowers the planner's threshold (read when the length is first planned: one process per variant)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
for prec, sizes in (("f32", [8000, 9216, 10000, 12000, 12288, 15625, 16000, 16807, 18000, 19683, 20480]),
                    ("f64", [5000, 6144, 8000, 9604, 10000, 10240])):
    for n in sizes:
        variants = (("lds", {}), ("hx", {"PFFT_JIT_HX_MIN_KIB": "48"}),
                    ("hx512", {"PFFT_JIT_HX_MIN_KIB": "48", "PFFT_JIT_HX_FORCE": "512"}))
        # (profiles/r5_perf_hx_below_pf512.txt also holds a fourth column, hx512pf: the software-pipelined form planned at
        # run time for 512 lanes -- a planner experiment that gained nothing and was removed again)
        for tag, env in variants:
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), tag], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, tag, p.stderr[-300:])), flush=True)
