"""Which split n1 x n2 of a four-step (GLOBAL tier) length is fastest: PFFT_GLOBAL_N1 forced over the divisors of n
against the planner's own choice.  usage: split_sweep.py <precision> <n> [<n> ...]  (1 GiB of data per length)"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(prec, n, batch, env):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--manual", "d=cpx,n=%d,b=%d" % (n, batch), "--precision", prec,
           "--no-cpu-baseline", "--steps", "40", "--warmup", "5"]
    p = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True)
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            return d["roofline"]["frac"], d["roofline"]["kernel"]
    return None, (p.stderr.strip().splitlines() or ["?"])[-1][:100]


def main():
    prec = sys.argv[1]
    esz = 8 if prec == "float" else 16
    for n in map(int, sys.argv[2:]):
        batch = max(1, (1 << 30) // (n * esz))
        base, label = run(prec, n, batch, {})
        print("n = %d batch %d: planner %s  %s" % (n, batch, base, label), flush=True)
        rows = []
        for c in range(32, 4097):
            if n % c or not (32 <= n // c <= 4096):
                continue
            f, label = run(prec, n, batch, {"PFFT_GLOBAL_N1": str(c)})
            rows.append((f or 0.0, c, n // c, label))
        rows.sort(reverse=True)
        for f, c, m, label in rows[:int(os.environ.get('SPLIT_TOP', '6'))]:
            print("   %4d x %4d  %.4f  %s" % (c, m, f, label), flush=True)


main()
