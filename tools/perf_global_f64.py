import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
# fp64 four-step sizes, 1 GiB per buffer
for lg in (14, 15, 16, 17, 18, 19, 20, 21, 22):
    run("f64 N=2^%d b=%d" % (lg, (1 << 26) >> lg), [1 << lg], (1 << 26) >> lg, "f64", reps=5)
run("f64 N=10^6 b=64", [1000000], 64, "f64", reps=5)
run("f64 N=30000 b=2048", [30000], 2048, "f64", reps=5)
