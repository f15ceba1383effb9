import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
# four-step sizes with one small odd factor (k * 2^m): about 1 GiB per buffer
for n in (3 << 16, 5 << 15, 3 << 18, 5 << 17, 7 << 17, 9 << 16, 15 << 16, 3 << 19, 5 << 18, 3 << 20):
    run("f32 N=%d" % n, [n], max(1, (128 << 20) // n), reps=5)
for n in (3 << 15, 3 << 18, 5 << 17, 3 << 19):
    run("f64 N=%d" % n, [n], max(1, (64 << 20) // n), "f64", reps=5)
