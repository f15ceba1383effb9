# round 5: grid rule of the one-per-CU register-resident kernels (persistent 2 x resident against k transforms per work-group)
mkdir -p gpurun_out/r5_run28
for c in g32_15 g64_14; do for g in default 8 4 2 1; do
  if [ $g = default ]; then unset PFFT_GROUPS_PER_WG; else export PFFT_GROUPS_PER_WG=$g; fi
  python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c gpw=$g', d['ms_per_step'], r['frac'], r.get('frac_wall'), r.get('kernel_ms_min_median_max'))"
done; done 2>&1 | tee gpurun_out/r5_run28/grid_rule_bench.txt
unset PFFT_GROUPS_PER_WG
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run28/grid_rule_jit.txt
import os, subprocess, sys
ROOT = os.getcwd()
CHILD = r"""
import sys; sys.path.insert(0, %r)
from perf_survey_lib import run
prec, n = sys.argv[1], int(sys.argv[2])
es = 8 if prec == "f32" else 16
run("%%s N=%%d %%s" %% (prec, n, sys.argv[3]), [n], max(1, (1 << 30) // (n * es)), prec)
""" % os.path.join(ROOT, "tools")
for prec, sizes in (("f32", [20480, 24576, 30000, 36864]), ("f64", [10752, 12288, 15000])):
    for n in sizes:
        for g in ("default", "8", "4", "2", "1"):
            e = dict(os.environ)
            if g != "default": e["PFFT_GROUPS_PER_WG"] = g
            p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), "gpw=" + g], env=e, capture_output=True, text=True)
            out = [l for l in p.stdout.splitlines() if "TB/s" in l]
            print(out[-1] if out else ("%s N=%d %s: failed %s" % (prec, n, g, p.stderr[-300:])), flush=True)
PY
