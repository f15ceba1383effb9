"""run one packed length a few times (for rocprofv3 counter passes): one_size.py N [prec] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perf_survey_lib import run
n = int(sys.argv[1]); prec = sys.argv[2] if len(sys.argv) > 2 else "f32"; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tot = (1 << 27) if prec == "f32" else (1 << 26)
run("%s N=%d" % (prec, n), [n], tot // n, prec, reps=reps)
