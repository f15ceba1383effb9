#!/bin/bash
# round 6, call 26: P -> BI / BI -> P at unaligned batch counts on default policies + shared walk (row-staged forms included),
# against the streamed twin; then the layout tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for env in "" "PFFT_NO_UNALIGNED_POLICY=1"; do
env $env python3 - "$env" <<'PY' >> gpurun_out/r6_pbi_policy.txt 2>&1
import sys; sys.path.insert(0, 'tools')
from perf_survey_lib import run
tag = "streamed" if sys.argv[1] else "policy3"
for prec, n, bs in (("f32", 1024, (131072, 131077, 33000)), ("f32", 256, (524293,)), ("f64", 1024, (65539,)), ("f32", 768, (174769,)), ("f32", 100, (1342181,))):
    for b in bs:
        run("%s %s N=%d b=%d P->BI" % (tag, prec, n, b), [n], b, prec, reps=5, backward_strides=[b], backward_distance=1)
        run("%s %s N=%d b=%d BI->P" % (tag, prec, n, b), [n], b, prec, reps=5, forward_strides=[b], forward_distance=1)
PY
done
grep -v amdgpu gpurun_out/r6_pbi_policy.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "strided or layouts or multiple_of_a_line or reference_size_grid or random_descriptors" 2>&1 | tail -3
python tools/fuzz.py 97 120 2>&1 | tail -2
