import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
run("f32 N=10^6 b=128", [1000000], 128, reps=5)
run("f32 N=62500 b=2048", [62500], 2048, reps=5)
run("f32 N=30000 b=4096", [30000], 4096, reps=5)
run("f32 N=12^6 b=40", [2985984], 40, reps=5)
run("f32 2D 1000x1000 b=128", [1000, 1000], 128, reps=5)
run("f32 BI N=1000", [1000], 131072, forward_strides=[131072], forward_distance=1, backward_strides=[131072], backward_distance=1)
