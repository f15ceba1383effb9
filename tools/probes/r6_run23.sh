#!/bin/bash
# round 6, call 23: the new test, N-D shapes with unaligned pitches (policy 3 A/B), BI survey at odd batch counts
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "multiple_of_a_line" 2>&1 | tail -3
python3 - <<'PY' > gpurun_out/r6_nd_unaligned.txt 2>&1
import os, sys, subprocess
CHILD = r"""
import sys; sys.path.insert(0, 'tools')
from perf_survey_lib import run
prec = sys.argv[1]; dims = [int(v) for v in sys.argv[2].split('x')]
n = 1
for d in dims: n *= d
es = 8 if prec == 'f32' else 16
run('%s %s %s' % (prec, sys.argv[2], sys.argv[3]), dims, max(1, (1 << 30) // (n * es)), prec, reps=5)
"""
for prec, shape in (("f32", "100x100x100"), ("f32", "200x300x500"), ("f32", "1000x1000"), ("f32", "1080x1920"), ("f32", "96x96x96"), ("f64", "100x100x100"), ("f32", "50x60x70x80"), ("f32", "360x360x360"), ("f32", "1000x30x30")):
    for tag, env in (("policy3", {}), ("streamed", {"PFFT_NO_UNALIGNED_POLICY": "1"})):
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, "-c", CHILD, prec, shape, tag], env=e, capture_output=True, text=True)
        print((p.stdout.strip().splitlines() or [p.stderr[-200:]])[-1], flush=True)
PY
cat gpurun_out/r6_nd_unaligned.txt
