import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
# long 1-D transforms: two stages up to 2^22, three stages from 2^23 (plan_global.cpp plan_three_stage); about 1 GiB per buffer
for e in (21, 22, 23, 24, 25, 26, 27):
    run("f32 N=2^%d" % e, [1 << e], max(1, (128 << 20) >> e), reps=5)
run("f32 N=3*2^22", [3 << 22], 10, reps=5)
run("f32 N=2^28 b=1", [1 << 28], 1, reps=3)
for e in (21, 22, 23, 24, 26):
    run("f64 N=2^%d" % e, [1 << e], max(1, (64 << 20) >> e), "f64", reps=5)
