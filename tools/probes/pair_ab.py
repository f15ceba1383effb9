import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from perf_survey_lib import run
import portfft_amd as pf
S = pf.complex_storage.SPLIT_COMPLEX
run("f32 N=2^22 b=32", [1 << 22], 32, reps=5)
run("f32 N=2^21 b=64", [1 << 21], 64, reps=5)
run("f32 N=2^20 b=128 split", [1 << 20], 128, complex_storage=S)
run("f64 N=2^20 b=64 split", [1 << 20], 64, "f64", complex_storage=S)
run("f32 N=65536 b=2Ki split", [65536], 2 << 10, complex_storage=S)
run("f32 BI N=2048 (col,col)", [2048], 65536, forward_strides=[65536], forward_distance=1, backward_strides=[65536], backward_distance=1)
