"""Register-resident packed kernel (stockham_wg_hx.hpp) at runtime-specialised lengths against the four-step plan of
the same descriptor (PFFT_NO_REGRES=1; the knobs are read at commit, so both run in one process).  ~1 GiB per buffer.
usage: perf_hx.py [f32|f64|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_survey_lib import run
which = sys.argv[1] if len(sys.argv) > 1 else "all"
F32 = [20736, 21504, 22528, 24576, 25000, 27000, 28672, 30000, 30720, 32000, 34560, 36864, 40000, 40960]
F64 = [10368, 10752, 12000, 12288, 14336, 15000, 16000, 18432, 20000, 20480]
for prec, sizes in (("f32", F32), ("f64", F64)):
    if which not in ("all", prec):
        continue
    es = 8 if prec == "f32" else 16
    for n in sizes:
        batch = max(1, (1 << 30) // (n * es))
        os.environ.pop("PFFT_NO_REGRES", None)
        run("%s N=%d hx" % (prec, n), [n], batch, prec)
        os.environ["PFFT_NO_REGRES"] = "1"
        run("%s N=%d four-step" % (prec, n), [n], batch, prec)
os.environ.pop("PFFT_NO_REGRES", None)
