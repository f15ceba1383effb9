"""N-D shapes in SPLIT_COMPLEX storage whose column passes have row pitches that are no multiple of a line: policy 3 (default since the end of
round 6) against the streamed twin (PFFT_NO_SPLIT_UNALIGNED_POLICY=1 in the environment)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from perf_survey_lib import run
import portfft_amd as pf
tag = "streamed" if os.environ.get("PFFT_NO_SPLIT_UNALIGNED_POLICY") == "1" else "policy 3"
for prec in ("f32", "f64"):
    k = 1 if prec == "f32" else 2
    for lengths, batch in (([1000, 1000], 128 // k), ([500, 1000], 256 // k), ([100, 100, 100], 128 // k), ([1000, 100], 1280 // k), ([360, 360], 1000 // k), ([1024, 1000], 128 // k)):
        run("%s %s %s split x%d" % (tag, prec, "x".join(map(str, lengths)), batch), lengths, batch, prec, reps=5, complex_storage=pf.complex_storage.SPLIT_COMPLEX)
