"""Grid rule (groups per work-group, PFFT_GROUPS_PER_WG) of the packed kernels and the 2-D passes, random data."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("f32", [16], 1 << 23), ("f32", [64], 1 << 21), ("f32", [256], 1 << 19), ("f32", [512], 1 << 18), ("f32", [1024], 1 << 17),
         ("f32", [2048], 1 << 16), ("f32", [4096], 1 << 15), ("f32", [8192], 1 << 14), ("f32", [16384], 1 << 13),
         ("f64", [256], 1 << 18), ("f64", [1024], 1 << 16), ("f64", [4096], 1 << 14), ("f32", [1024, 1024], 128), ("f64", [1024, 1024], 64)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, HERE)
    import perf_survey_lib as L
    prec, dims, batch = sys.argv[2], [int(x) for x in sys.argv[3].split("x")], int(sys.argv[4])
    L.run("%s %s x %d gpw=%s" % (prec, sys.argv[3], batch, os.environ.get("PFFT_GROUPS_PER_WG", "entry")), dims, batch, prec, reps=20)
    sys.exit(0)
for prec, dims, batch in CASES:
    for g in ("", "1", "2", "4", "8", "0"):
        env = dict(os.environ)
        if g: env["PFFT_GROUPS_PER_WG"] = g
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", prec, "x".join(map(str, dims)), str(batch)], env=env, capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if "TB/s" in l]
        print(out[-1] if out else "FAILED %s: %s" % (dims, (p.stderr or p.stdout)[-300:]), flush=True)
