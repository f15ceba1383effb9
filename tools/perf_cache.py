"""Cache-sized chunks + writer/reader cache policies (plan_core.cpp cache_chunk_bytes) against everything-streamed, random
data, through the library: small and large batches of the two-launch plans.  Each case runs in a child process with
PFFT_CACHE_CHUNK_MIB unset (256) / 0 / other values."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("f32", [1024, 1024], 4), ("f32", [1024, 1024], 16), ("f32", [1024, 1024], 32), ("f32", [1024, 1024], 64), ("f32", [1024, 1024], 256),
         ("f32", [2048, 2048], 8), ("f32", [512, 512], 128), ("f64", [1 << 20], 4), ("f64", [1 << 20], 8), ("f64", [1 << 20], 16),
         ("f64", [1 << 20], 128), ("f32", [65536], 256), ("f32", [65536], 2048), ("f32", [1 << 20], 32), ("f32", [1 << 20], 256),
         ("f64", [65536], 512)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, HERE)
    import perf_survey_lib as L
    prec, dims, batch = sys.argv[2], [int(x) for x in sys.argv[3].split("x")], int(sys.argv[4])
    L.run("%s %s x %d chunk=%s" % (prec, sys.argv[3], batch, os.environ.get("PFFT_CACHE_CHUNK_MIB", "256(default)")), dims, batch, prec, reps=20)
    sys.exit(0)
settings = sys.argv[1:] or ["", "0"]
for prec, dims, batch in CASES:
    for setting in settings:
        env = dict(os.environ)
        if setting != "":
            env["PFFT_CACHE_CHUNK_MIB"] = setting
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", prec, "x".join(map(str, dims)), str(batch)], env=env, capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if "TB/s" in l]
        print(out[-1] if out else "FAILED %s %s: %s" % (prec, dims, (p.stderr or p.stdout)[-300:]), flush=True)
