"""P -> BI and BI -> P at lengths whose full-width group is beyond the LDS (1025 ... 2048 points): what the row-staged forms deliver."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from perf_survey_lib import run
for prec, n, b in (("f32", 2048, 65600), ("f32", 1280, 104832), ("f32", 1024, 131136), ("f64", 2048, 32832), ("f64", 1280, 52416), ("f64", 1024, 65600)):
    bi = dict(forward_strides=[b], forward_distance=1, backward_strides=[b], backward_distance=1)
    run("%s N=%d BI->BI (col,col)" % (prec, n), [n], b, prec, reps=5, **bi)
    run("%s N=%d P->BI (row,col)" % (prec, n), [n], b, prec, reps=5, backward_strides=[b], backward_distance=1)
    run("%s N=%d BI->P (col,row)" % (prec, n), [n], b, prec, reps=5, forward_strides=[b], forward_distance=1)
    run("%s N=%d P->P (spec)" % (prec, n), [n], b, prec, reps=5)
