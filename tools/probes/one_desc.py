"""run one 1-D descriptor a few times (for rocprofv3 passes): one_desc.py N batch [prec] [reps] [bi]
(`bi`: batch-interleaved on both sides instead of packed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perf_survey_lib import run
n, batch = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
kw = {}
if len(sys.argv) > 5 and sys.argv[5] == "bi":
    kw = dict(forward_strides=[batch], forward_distance=1, backward_strides=[batch], backward_distance=1)
run("%s N=%d b=%d%s" % (prec, n, batch, " BI" if kw else ""), [n], batch, prec, reps=reps, **kw)
