"""radix sequences (and lanes per FFT) of the runtime planner for the long single-work-group lengths, forced through
PFFT_JIT_SPEC_RADICES / PFFT_JIT_FORCE_TPF (one process per point: the kernel cache is per process)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
cases = {
    15625: [("5x5x5x5x5x5", None), ("25x25x25", None), ("25x25x25", 625), ("25x25x5x5", None), ("25x5x5x5x5", None), ("25x25x25", 313)],
    3125: [("5x5x5x5x5", None), ("25x25x5", None), ("25x5x25", None), ("5x25x25", None), ("25x25x5", 125)],
    6561: [("9x9x9x9", None), ("27x27x9", None), ("27x9x27", None), ("9x27x27", None), ("27x27x9", 243)],
    2187: [("9x9x9x3", None), ("27x9x9", None), ("27x27x3", None), ("9x9x27", None)],
    2401: [("7x7x7x7", None), ("49x49", None), ("49x7x7", None)],
    16807: [("7x7x7x7x7", None), ("49x49x7", None), ("49x7x49", None), ("7x49x49", None)],
    19683: [("9x9x9x9x3", None), ("27x27x27", None), ("27x27x27", 729)],
    3000: [("20x15x10", None), ("10x10x10x3", None), ("15x20x10", None), ("10x15x20", None), ("12x10x5x5", None), ("30x10x10", None)],
    6000: [("10x10x10x6", None), ("20x20x15", None), ("20x15x20", None), ("15x20x20", None), ("30x20x10", None), ("24x25x10", None)],
    10080: [("12x12x10x7", None), ("16x15x7x6", None), ("21x20x24", None), ("20x21x24", None), ("14x12x10x6", None), ("18x16x7x5", None)],
    7680: [("12x10x8x8", None), ("16x16x30", None), ("32x16x15", None), ("16x15x32", None), ("16x16x6x5", None), ("20x16x24", None)],
    625: [("5x5x5x5", None), ("25x25", None)],
}
only = [int(a) for a in sys.argv[1:]]
for n, variants in cases.items():
    if only and n not in only:
        continue
    for rad, tpf in variants:
        env = dict(os.environ, PFFT_NO_PRECOMPILED="1", PFFT_JIT_SPEC_RADICES="%d:%s" % (n, rad))
        if tpf:
            env["PFFT_JIT_FORCE_TPF"] = str(tpf)
        r = subprocess.run([sys.executable, os.path.join(HERE, "probes", "one_size.py"), str(n), "f32", "10"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if "TB/s" in l]
        print("N=%-6d %-16s T=%-5s %s" % (n, rad, tpf or "auto", line[0][44:90] if line else "FAIL " + r.stderr[-300:].replace("\n", " ")), flush=True)
