import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
import portfft_amd as pf
S = pf.complex_storage.SPLIT_COMPLEX
run("f32 N=4096 b=64Ki split", [4096], 64 << 10, complex_storage=S)
run("f32 N=256 b=512Ki split", [256], 512 << 10, complex_storage=S)
run("f32 N=16 b=8Mi split", [16], 8 << 20, complex_storage=S)
run("f64 N=1024 b=128Ki split", [1024], 128 << 10, "f64", complex_storage=S)
run("f32 2D 1024x1024 b=256 split", [1024, 1024], 256, complex_storage=S, reps=3)
