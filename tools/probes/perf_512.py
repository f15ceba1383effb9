import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
run("f32 N=2^18 b=512", [1 << 18], 512, reps=10)
run("f32 2D 512x512 b=512", [512, 512], 512, reps=10)
run("f32 BI N=512", [512], 262144, forward_strides=[262144], forward_distance=1, backward_strides=[262144], backward_distance=1)
run("f32 P->BI N=512", [512], 262144, backward_strides=[262144], backward_distance=1)
run("f32 BI->P N=512", [512], 262144, forward_strides=[262144], forward_distance=1)
run("f32 3D 512^3 b=1", [512, 512, 512], 1, reps=5)
