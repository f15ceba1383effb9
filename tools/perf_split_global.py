import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
import portfft_amd as pf
S = pf.complex_storage.SPLIT_COMPLEX
run("f32 N=65536 b=2Ki interleaved", [65536], 2 << 10)
run("f32 N=65536 b=2Ki split", [65536], 2 << 10, complex_storage=S)
run("f32 N=2^20 b=128 split", [1 << 20], 128, complex_storage=S)
run("f64 N=2^20 b=64 split", [1 << 20], 64, "f64", complex_storage=S)
run("f32 N=10^6 b=128 split", [1000000], 128, complex_storage=S)
run("f32 2D 1024x1024 b=128 split", [1024, 1024], 128, complex_storage=S)
