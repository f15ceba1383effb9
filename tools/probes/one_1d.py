"""one 1-D packed transform timed in a loop (for rocprofv3): one_1d.py <f32|f64> <n> <batch> [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from perf_survey_lib import run
prec, n, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
run("%s N=%d b=%d" % (prec, n, batch), [n], batch, prec, reps=int(sys.argv[4]) if len(sys.argv) > 4 else 10)
