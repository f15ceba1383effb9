#!/bin/bash
# round 6, call 25: P -> BI and BI -> P (row-staged forms) at aligned / unaligned batch counts
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 - <<'PY' > gpurun_out/r6_pbi_unaligned.txt 2>&1
import sys; sys.path.insert(0, 'tools')
from perf_survey_lib import run
for prec, n, bs in (("f32", 1024, (131072, 131077, 33000)), ("f32", 256, (524288, 524293)), ("f64", 1024, (65536, 65539)), ("f32", 768, (174768, 174769)), ("f32", 4096, (32768, 32771))):
    for b in bs:
        run("%s N=%d b=%d P->BI" % (prec, n, b), [n], b, prec, reps=5, backward_strides=[b], backward_distance=1)
        run("%s N=%d b=%d BI->P" % (prec, n, b), [n], b, prec, reps=5, forward_strides=[b], forward_distance=1)
PY
grep -v amdgpu gpurun_out/r6_pbi_unaligned.txt
