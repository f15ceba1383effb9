import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
for prec, n, b in (("f64", 1024, 128 << 10), ("f32", 1024, 256 << 10), ("f32", 256, 1 << 20), ("f64", 256, 512 << 10)):
    bi = dict(forward_strides=[b], forward_distance=1, backward_strides=[b], backward_distance=1)
    run("%s N=%d BI->BI (col,col)" % (prec, n), [n], b, prec, reps=5, **bi)
    run("%s N=%d P->BI (row,col)" % (prec, n), [n], b, prec, reps=5, backward_strides=[b], backward_distance=1)
    run("%s N=%d BI->P (col,row)" % (prec, n), [n], b, prec, reps=5, forward_strides=[b], forward_distance=1)
    run("%s N=%d P->P (spec)" % (prec, n), [n], b, prec, reps=5)
