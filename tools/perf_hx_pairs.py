"""Transforms of 64 ... 160 KiB (one LDS-resident work-group per CU today): the register-resident kernel planned as TWO
work-groups per CU (choose_hx_params, lanes <= 512, half image <= 80 KiB each; PFFT_JIT_HX_PAIR_MIN_KIB lowers the planner's
threshold) against the LDS-resident plan.  The threshold is read when the length is first planned: one process per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
F32 = [10500, 10935, 11000, 11250, 11520, 11664, 12150, 12500, 13000, 13125, 13500, 14000, 14580, 15000, 15120, 15625, 16000, 16200]
F64 = [5250, 5400, 5500, 5625, 5832, 6000, 6250, 6480, 6561, 6750, 7000, 7290, 7500, 7776, 8000]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for prec, sizes in (("f32", F32), ("f64", F64)):
    if which not in ("all", prec):
        continue
    for n in sizes:
        # lds: the LDS-resident plan (registered entry / tuned table / static rule); default: what a commit takes today;
        # pair: the planner's two-per-CU plan whatever the table or the registry hold for the length
        variants = (("lds", {"PFFT_JIT_HX_PAIRS": "0", "PFFT_NO_REGRES": "1"}), ("default", {}))
        if len(sys.argv) > 2 and sys.argv[2] == "forced":
            variants += (("pair", {"PFFT_JIT_HX_PAIR_MIN_KIB": "60", "PFFT_NO_PRECOMPILED": "1", "PFFT_NO_TUNED_TABLE": "1"}),)
        for tag, env in variants:
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), tag], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, tag, p.stderr[-300:])), flush=True)
