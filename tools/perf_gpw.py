"""Grid rule of the four-step stages: groups per work-group (PFFT_GROUPS_PER_WG) for several GLOBAL-tier sizes."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("f64", [1 << 20], 128), ("f64", [1 << 18], 512), ("f64", [1 << 22], 32), ("f64", [65536], 2048),
         ("f32", [1 << 20], 256), ("f32", [65536], 2048), ("f32", [1 << 22], 64), ("f32", [1 << 18], 1024)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, HERE)
    import perf_survey_lib as L
    prec, n, batch = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    L.run("%s N=%d x %d gpw=%s" % (prec, n, batch, os.environ.get("PFFT_GROUPS_PER_WG", "entry")), [n], batch, prec, reps=15)
    sys.exit(0)
for prec, dims, batch in CASES:
    for g in ("", "0", "2", "4", "8"):
        env = dict(os.environ)
        if g: env["PFFT_GROUPS_PER_WG"] = g
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", prec, str(dims[0]), str(batch)], env=env, capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if "TB/s" in l]
        print(out[-1] if out else "FAILED %s: %s" % (dims, (p.stderr or p.stdout)[-300:]), flush=True)
