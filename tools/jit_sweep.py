"""sweep lanes-per-FFT and padding of the runtime planner for a few lengths (one process per point: the kernel cache
is per process)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sizes = {384: [24, 32, 48, 64, 96], 320: [20, 32, 40, 64], 1280: [80, 128, 160, 256], 5120: [256, 320, 512],
         1200: [64, 100, 120, 128, 256], 3000: [128, 150, 200, 256, 300], 6000: [256, 300, 512, 600], 1920: [120, 128, 192, 256]}
for n, ts in sizes.items():
    for t in ts:
        for pad in (0, 16):
            env = dict(os.environ, PFFT_NO_PRECOMPILED="1", PFFT_JIT_FORCE_TPF=str(t), PFFT_JIT_FORCE_PAD=str(pad))
            r = subprocess.run([sys.executable, os.path.join(HERE, "one_size.py"), str(n), "f32", "10"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if "TB/s" in l]
            print("N=%-5d T=%-4d pad=%-2d %s" % (n, t, pad, line[0][44:75] if line else "FAIL " + r.stderr[-200:]), flush=True)
