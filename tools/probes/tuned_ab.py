"""Every entry of portfft_amd/csrc/tuned_gfx950.inc at 1 GiB per buffer through bench.py: the table's choice against the
static rule (PFFT_NO_TUNED_TABLE=1).  The table was measured on 256 MiB; this is the check at the bench's size."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(prec, n, env):
    es = 8 if prec == "float" else 16
    batch = max(1, (1 << 30) // (n * es))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--manual", "d=cpx,n=%d,b=%d" % (n, batch), "--precision", prec,
           "--no-cpu-baseline", "--steps", "40", "--warmup", "5"]
    p = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True)
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            return d["roofline"]["frac"], re.search(r"radices/factors (\S+)", d["roofline"]["kernel"]).group(1)
    return 0.0, "?"

for ln in open(os.path.join(ROOT, "portfft_amd", "csrc", "tuned_gfx950.inc")):
    m = re.match(r"\{PFFT_PRECISION_(F32|F64), (\d+),", ln)
    if not m:
        continue
    prec, n = ("float" if m.group(1) == "F32" else "double"), int(m.group(2))
    if os.environ.get("TUNED_AB_ONLY") and str(n) not in os.environ["TUNED_AB_ONLY"].split(","):
        continue
    a, ka = run(prec, n, {})
    b, kb = run(prec, n, {"PFFT_NO_TUNED_TABLE": "1"})
    print("%s n=%d: table %.4f (%s)  static %.4f (%s)  %+.1f %%" % (prec, n, a, ka, b, kb, 100 * (a / b - 1) if b else 0), flush=True)
