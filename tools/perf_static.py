import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
tot = 1 << 27
for n in (32, 64, 128, 256, 1024, 96, 192, 384, 768, 1536, 3072, 6144, 12288, 80, 100, 160, 320, 640, 1280, 2560, 5120, 1000, 10000):
    run("f32 N=%d" % n, [n], tot // n, reps=10)
for n in (64, 256, 512, 1024, 2048, 8192, 96, 384, 1536, 6144, 100, 1000):
    run("f64 N=%d" % n, [n], (tot // 2) // n, "f64", reps=10)
