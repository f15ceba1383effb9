import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
run("f32 N=65536 b=2Ki", [65536], 2 << 10, reps=5)
run("f32 N=2^18 b=512", [1 << 18], 512, reps=5)
run("f32 N=2^20 b=128", [1 << 20], 128, reps=5)
run("f32 N=2^22 b=32", [1 << 22], 32, reps=5)
run("f32 N=10^6 b=128", [1000000], 128, reps=5)
run("f32 N=62500 b=2048", [62500], 2048, reps=5)
run("f32 N=30000 b=4096", [30000], 4096, reps=5)
run("f32 P->BI N=1024", [1024], 131072, backward_strides=[131072], backward_distance=1)
run("f32 BI->P N=1024", [1024], 131072, forward_strides=[131072], forward_distance=1)
run("f32 P->BI N=256", [256], 524288, backward_strides=[524288], backward_distance=1)
