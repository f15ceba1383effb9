import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
for n in (243, 625, 720, 1080, 1200, 2187, 2401, 3000, 3125, 4000, 4800, 6000, 7680, 10080):
    run("f32 N=%d" % n, [n], (1 << 27) // n, reps=10)
for n in (625, 720, 2187, 3000, 5040, 6561):
    run("f64 N=%d" % n, [n], (1 << 26) // n, "f64", reps=10)
