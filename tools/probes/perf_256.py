import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
run("f32 N=65536 b=2Ki", [65536], 2 << 10, reps=10)
run("f32 2D 256x256 b=2Ki", [256, 256], 2 << 10, reps=10)
run("f32 BI N=256", [256], 524288, forward_strides=[524288], forward_distance=1, backward_strides=[524288], backward_distance=1)
run("f32 P->BI N=256", [256], 524288, backward_strides=[524288], backward_distance=1)
run("f32 3D 256^3 b=8", [256, 256, 256], 8, reps=5)
