import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
for b in (8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 128, 192, 256):
    run("f32 N=4096 b=%dKi" % b, [4096], b << 10, reps=20)
