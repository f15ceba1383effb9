# round 5: the whole GPU suite on the final build, smoke, fuzz (general / pairs / regres), the second-choice pair plans against LDS
mkdir -p gpurun_out/r5_final4
( time timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 ) 2>&1 | tee gpurun_out/r5_final4/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r5_final4/smoke.txt
python tools/fuzz.py 67 150 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_final4/fuzz_67_150.txt
python tools/fuzz.py 74 150 pairs 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_final4/fuzz_74_150_pairs.txt
python tools/fuzz.py 68 60 regres 2>&1 | grep -v "^\.\.\.\|^skip\|amdgpu" | tee gpurun_out/r5_final4/fuzz_68_60_regres.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_final4/perf_second_choice_pairs.txt
import os, subprocess, sys
ROOT = os.getcwd()
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
for prec, sizes in (("f32", [11664, 10935, 12960, 11200]), ("f64", [7168, 6656, 6480, 7200, 5760, 7680])):
    for n in sizes:
        for tag, env in (("lds", {"PFFT_JIT_HX_PAIRS": "0", "PFFT_NO_REGRES": "1"}), ("default", {})):
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), tag], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, tag, p.stderr[-300:])), flush=True)
PY
python bench.py 2>/dev/null | tail -1 > gpurun_out/r5_final4/bench_default.json
