import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
for prec in ("f32", "f64"):
    b = (32 << 10) if prec == "f32" else (16 << 10)
    run(prec + " N=4096 packed", [4096], b, prec)
    run(prec + " N=4096 rows of a padded matrix (ld=4160)", [4096], b, prec, forward_distance=4160, backward_distance=4160)
    run(prec + " N=4096 padded in, packed out", [4096], b, prec, forward_distance=4160)
    run(prec + " N=4096 every 2nd sample", [4096], b, prec, forward_strides=[2], forward_distance=8192, backward_strides=[2], backward_distance=8192)
    run(prec + " N=1000 ld=1024", [1000], b * 4, prec, forward_distance=1024, backward_distance=1024)
    run(prec + " N=1200 ld=1280 (jit)", [1200], b * 3, prec, forward_distance=1280, backward_distance=1280)
    run(prec + " N=64 ld=80", [64], b * 64, prec, forward_distance=80, backward_distance=80)
    run(prec + " N=16 ld=20", [16], b * 256, prec, forward_distance=20, backward_distance=20)
