"""Split entries of a freshly generated table that the library on this box does not have yet: PFFT_GLOBAL_N1=<n1> against
the library's own choice, 1 GiB per buffer.  usage: split_ab.py <table.inc> <n,n,...>"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
only = set(sys.argv[2].split(","))

def run(prec, n, env):
    es = 8 if prec == "float" else 16
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--manual", "d=cpx,n=%d,b=%d" % (n, max(1, (1 << 30) // (n * es))),
           "--precision", prec, "--no-cpu-baseline", "--steps", "30", "--warmup", "3"]
    p = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True)
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            return d["roofline"]["frac"], re.search(r"radices/factors (\S+)", d["roofline"]["kernel"]).group(1)
    return 0.0, "?"

for ln in open(sys.argv[1]):
    m = re.match(r"\{PFFT_PRECISION_(F32|F64), (\d+), 1, 2, \{(\d+), (\d+)\}\}", ln)
    if not m or m.group(2) not in only:
        continue
    prec, n, n1 = ("float" if m.group(1) == "F32" else "double"), int(m.group(2)), m.group(3)
    a, ka = run(prec, n, {"PFFT_GLOBAL_N1": n1})
    b, kb = run(prec, n, {})
    print("%s n=%d: table %.4f (%s)  static %.4f (%s)  %+.1f %%" % (prec, n, a, ka, b, kb, 100 * (a / b - 1) if b else 0), flush=True)
