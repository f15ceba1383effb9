import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import portfft_amd as pf
from perf_survey_lib import run
run("f64 N=2^20 b=128 (C3)", [1 << 20], 128, "f64", reps=5)
run("f32 N=65536 b=2Ki", [65536], 2 << 10, reps=5)
run("f32 N=2^20 b=256", [1 << 20], 256, reps=5)
