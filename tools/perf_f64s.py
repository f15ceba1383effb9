import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
run("f64 N=65536 b=1Ki", [65536], 1 << 10, "f64", reps=10)
run("f64 N=2^18 b=256", [1 << 18], 256, "f64", reps=10)
run("f64 2D 256x256 b=1Ki", [256, 256], 1 << 10, "f64", reps=10)
run("f64 2D 512x512 b=256", [512, 512], 256, "f64", reps=10)
run("f64 BI N=256", [256], 262144, "f64", forward_strides=[262144], forward_distance=1, backward_strides=[262144], backward_distance=1)
run("f64 BI N=512", [512], 131072, "f64", forward_strides=[131072], forward_distance=1, backward_strides=[131072], backward_distance=1)
run("f64 3D 256^3 b=4", [256, 256, 256], 4, "f64", reps=5)
