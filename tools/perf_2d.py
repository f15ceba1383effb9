"""Two-pass 2-D plan (stockham_rows2d.hpp) against rows-then-full-columns, ~1 GiB per buffer, several shapes.
usage: perf_2d.py   (runs each shape in a child process with PFFT_2D_TWO_PASS=1 / 0)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = [("f32", [256, 256]), ("f32", [512, 512]), ("f32", [1024, 1024]), ("f32", [2048, 2048]), ("f32", [4096, 4096]),
          ("f32", [1000, 1000]), ("f32", [768, 768]), ("f32", [1200, 1200]), ("f32", [1536, 1536]), ("f32", [3000, 1000]),
          ("f32", [64, 1024]), ("f32", [16, 2048]), ("f32", [1024, 256]), ("f32", [4096, 128]), ("f32", [128, 4096]),
          ("f32", [8, 512, 512]), ("f32", [256, 256, 256]),
          ("f64", [1024, 1024]), ("f64", [512, 512]), ("f64", [1000, 1000]), ("f64", [2048, 2048]), ("f64", [256, 256])]
if len(sys.argv) > 1 and sys.argv[1] == "more":
    SHAPES = [("f32", [384, 384]), ("f32", [640, 640]), ("f32", [896, 896]), ("f32", [960, 960]), ("f32", [1280, 1280]),
              ("f32", [1080, 1920]), ("f32", [1920, 1080]), ("f32", [720, 1280]), ("f32", [480, 640]), ("f32", [2000, 2000]),
              ("f32", [3072, 3072]), ("f32", [8192, 8192]), ("f32", [100, 10000]), ("f64", [768, 768]), ("f64", [1536, 1536]),
              ("f64", [4096, 4096]), ("f64", [1080, 1920])]
if len(sys.argv) > 1 and sys.argv[1] == "jit":
    SHAPES = [("f32", [384, 384]), ("f32", [768, 768]), ("f32", [896, 896]), ("f32", [960, 960]), ("f32", [1000, 1000]),
              ("f32", [1200, 1200]), ("f32", [1280, 1280]), ("f32", [1536, 1536]), ("f32", [1080, 1920]), ("f32", [1920, 1080]),
              ("f32", [2000, 2000]), ("f32", [3072, 3072]), ("f32", [3000, 1000]), ("f32", [4096, 4096]), ("f32", [4096, 128]),
              ("f64", [768, 768]), ("f64", [1000, 1000]), ("f64", [1536, 1536]), ("f64", [1080, 1920]), ("f64", [4096, 4096])]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, HERE)
    import perf_survey_lib as L
    prec, dims = sys.argv[2], [int(x) for x in sys.argv[3].split("x")]
    n = 1
    for d in dims: n *= d
    batch = max(1, (1 << 30) // (n * (8 if prec == "f32" else 16)))
    L.run("%s %s x %d %s" % (prec, sys.argv[3], batch, "two-pass" if os.environ.get("PFFT_2D_TWO_PASS") != "0" else "rows+cols"), dims, batch, prec)
    sys.exit(0)
for prec, dims in SHAPES:
    for mode in ("1", "0"):
        env = dict(os.environ, PFFT_2D_TWO_PASS=mode)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", prec, "x".join(map(str, dims))], env=env, capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if "TB/s" in l]
        print(out[-1] if out else "FAILED %s %s mode %s: %s" % (prec, dims, mode, (p.stderr or p.stdout)[-300:]), flush=True)
