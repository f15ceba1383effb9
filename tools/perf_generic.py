import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
tot = 1 << 28   # elements per buffer (2 GiB fp32)
for n in (120, 243, 343, 625, 720, 1080, 1200, 1920, 2187, 2401, 3000, 3125, 4000, 4800, 6000, 6561, 7680, 10080, 15625, 16807, 18000, 19683, 20480):
    run("f32 N=%d" % n, [n], tot // n)
for n in (1200, 2187, 3000, 5040, 9604, 10125, 10240):
    run("f64 N=%d" % n, [n], (tot // 2) // n, "f64")
