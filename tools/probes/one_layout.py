"""one 1-D transform with a chosen layout pair timed in a loop (for rocprofv3): one_layout.py <f32|f64> <n> <batch> <P|BI> <P|BI>"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from perf_survey_lib import run
prec, n, b, li, lo = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
kw = {}
if li == "BI":
    kw.update(forward_strides=[b], forward_distance=1)
if lo == "BI":
    kw.update(backward_strides=[b], backward_distance=1)
run("%s N=%d %s->%s" % (prec, n, li, lo), [n], b, prec, reps=5, **kw)
