# round 5: the pair threshold between 74 and 80 KiB (the reference's regression size 9800 is 76.6 KiB)
mkdir -p gpurun_out/r5_run37
for v in default pair74; do
  if [ $v = pair74 ]; then export PFFT_JIT_HX_PAIR_MIN_KIB=74 PFFT_NO_TUNED_TABLE=1; fi
  python bench.py --config ref9800 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('ref9800 $v', d['ms_per_step'], r['frac'], r.get('frac_wall'), r['kernel'])"
done 2>&1 | tee gpurun_out/r5_run37/ref9800.txt
unset PFFT_JIT_HX_PAIR_MIN_KIB PFFT_NO_TUNED_TABLE
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run37/perf_pairs_74_80.txt
import os, subprocess, sys
ROOT = os.getcwd()
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
for prec, sizes in (("f32", [9600, 9728, 9800, 9984, 10000, 10080, 10240]), ("f64", [4800, 4864, 4900, 5000, 5040, 5120])):
    for n in sizes:
        for tag, env in (("default", {}), ("pair from 74 KiB", {"PFFT_JIT_HX_PAIR_MIN_KIB": "74", "PFFT_NO_TUNED_TABLE": "1", "PFFT_NO_PRECOMPILED": "1"})):
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), tag], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, tag, p.stderr[-300:])), flush=True)
PY
