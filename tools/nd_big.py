import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
run("f32 128x128 b=8192", [128, 128], 8192)
run("f32 128x128x128 b=64", [128, 128, 128], 64)
run("f32 64x256 b=8192", [64, 256], 8192)
run("f64 64x128 b=8192", [64, 128], 8192, "f64")
run("f64 128x128 b=2048", [128, 128], 2048, "f64")
run("f32 256x256x256 b=8", [256, 256, 256], 8)
