"""tools/perf_generic.py rows 3000 ... 10080 with and without PFFT_PLAN_MEASURE=1 (a process per row and mode: the
kernel tables are per process).  usage: perf_generic_measured.py [cache_dir]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
cache = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pfft_measure_cache"
rows = [("f32", n) for n in (1080, 1200, 1920, 2187, 2401, 3000, 3125, 4000, 4800, 6000, 6561, 7680, 10080)] + \
       [("f64", n) for n in (1200, 2187, 3000, 5040)]
for prec, n in rows:
    line = "%s N=%-6d" % (prec, n)
    for mode in ("0", "1"):
        env = dict(os.environ, PFFT_JIT_CACHE_DIR=cache, PFFT_PLAN_MEASURE=mode)
        r = subprocess.run([sys.executable, os.path.join(HERE, "probes", "one_size.py"), str(n), prec, "10"], env=env,
                           capture_output=True, text=True)
        got = [l for l in r.stdout.splitlines() if "TB/s" in l]
        line += "  %s: %s" % ("measured" if mode == "1" else "static  ", got[0][44:100] if got else "FAIL " + r.stderr[-200:].replace("\n", " "))
    print(line, flush=True)
