import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
tot = 1 << 27
for n in (37, 61, 37 * 64, 61 * 16, 43 * 47, 59 * 59):
    run("f32 N=%d" % n, [n], tot // n)
run("f64 N=2368", [37 * 64], (tot // 2) // (37 * 64), "f64")
