"""Four-step lengths whose stage B reads a row-major intermediate (no tiled form: n2 or its first pass is not a multiple of
stage A's group width): the row-lanes form of the runtime-specialised stage B (strided_pass TIN = -1: 64 consecutive elements
of one row per wave) against its f-fastest form (PFFT_NO_TIN_ROWS=1) and, where rows are short enough to be staged through
LDS by default, against that (PFFT_ROW_IN_MAX_N=0 takes the staging away so that the row-lanes form runs).  ~1 GiB per buffer.
The knobs are read at commit, so all variants run in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
F32 = [1000000, 68640, 250000, 62500, 100000, 120000, 500000, 2985984, 390625, 531441, 48000, 51200, 60000, 160000]
F64 = [1000000, 68640, 250000, 62500, 100000, 30000, 50000]
KN = ("PFFT_TIN_ROWS", "PFFT_ROW_IN_MAX_N")
for prec, sizes in (("f32", F32), ("f64", F64)):
    es = 8 if prec == "f32" else 16
    for n in sizes:
        batch = max(1, (1 << 30) // (n * es))
        for tag, env in (("default", {}), ("rows", {"PFFT_TIN_ROWS": "1"}),
                         ("rows, no staging", {"PFFT_TIN_ROWS": "1", "PFFT_ROW_IN_MAX_N": "0"}),
                         ("f-fastest, no staging", {"PFFT_ROW_IN_MAX_N": "0"})):
            for k in KN:
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                run("%s N=%d %s" % (prec, n, tag), [n], batch, prec)
            except Exception as e:  # noqa: BLE001
                print("%s N=%d %s: %r" % (prec, n, tag, e), flush=True)
for k in KN:
    os.environ.pop(k, None)
