"""BATCH_INTERLEAVED at realistic batch counts (VERDICT r5 task 5): the survey's BI rows use a power-of-two batch, i.e. an element
stride of 1-2 MiB -- every butterfly leg of a work-group lands on the same few HBM channels.  N = 256 / 1024 / 4096, fp32 and
fp64, batch = 2^17, 2^17 +- 64, 100 000, 99 968 (a multiple of 64 that is no power of two) and 33 000 (the reference's test
batch): BI -> BI, and the packed transform of the same size beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
for prec in ("f32", "f64"):
    for n in (256, 1024, 4096):
        scale = (4096 // n) * (1 if prec == "f32" else 1)
        for b0 in (131072, 131072 + 64, 131072 - 64, 100000, 99968, 33000):
            b = b0 * max(1, scale // (2 if prec == "f64" else 1)) if n < 4096 else b0 // (2 if prec == "f64" else 1)
            if n < 4096 and b0 in (100000, 33000):
                b = b0 * max(1, scale // (2 if prec == "f64" else 1))
            bi = dict(forward_strides=[b], forward_distance=1, backward_strides=[b], backward_distance=1)
            try:
                run("%s N=%d batch %d BI->BI" % (prec, n, b), [n], b, prec, reps=5, **bi)
            except Exception as e:  # noqa: BLE001
                print("%s N=%d batch %d: %r" % (prec, n, b, e), flush=True)
        run("%s N=%d batch %d P->P" % (prec, n, b), [n], b, prec, reps=5)
