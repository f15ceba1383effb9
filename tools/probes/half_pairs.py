"""four-step sizes k * 2^m: half pairs (registered stage B + runtime-specialised stage A of the same group width)
against the round's earlier plan (PFFT_NO_HALF_PAIRS=1, or the settings in AB_ENV=K=V,K=V) -- error against torch.fft and time, in child processes"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, os, torch
sys.path.insert(0, os.path.join(%r, "..", ".."))
import portfft_amd as pf
prec, n, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ct = torch.complex64 if prec == "f32" else torch.complex128
d = pf.descriptor([n], prec); d.number_of_transforms = batch; d.placement = pf.placement.OUT_OF_PLACE
plan = d.commit()
x = torch.empty(batch * n, dtype=ct, device="cuda"); torch.view_as_real(x).uniform_(-1, 1)
y = torch.empty_like(x)
plan.compute_forward(x, y); torch.cuda.synchronize()
nb = min(batch, 4)
ref = torch.fft.fft(x.view(batch, n)[:nb].to(torch.complex128), dim=1)
err = ((y.view(batch, n)[:nb].to(torch.complex128) - ref).norm() / ref.norm()).item()
z = torch.empty_like(x)
plan.compute_backward(y, z); torch.cuda.synchronize()
errb = ((z.view(batch, n)[:nb] / n - x.view(batch, n)[:nb]).norm() / x.view(batch, n)[:nb].norm()).item()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): plan.compute_forward(x, y)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("%%s N=%%d b=%%d: %%.4f ms  %%.3f of peak  err %%.2e  roundtrip %%.2e" %% (prec, n, batch, ms, 2 * x.numel() * x.element_size() / (ms * 1e-3) / 8e12, err, errb))
''' % HERE
cases = [("f32", 3 << 16), ("f32", 5 << 15), ("f32", 3 << 18), ("f32", 5 << 17), ("f32", 7 << 17), ("f32", 15 << 16),
         ("f32", 9 << 16), ("f32", 3 << 15), ("f64", 3 << 15), ("f64", 3 << 18), ("f64", 5 << 17), ("f64", 7 << 15)]
if len(sys.argv) > 1:
    cases = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
for prec, n in cases:
    batch = max(1, ((128 << 20) if prec == "f32" else (64 << 20)) // n)
    other = dict(kv.split("=") for kv in os.environ.get("AB_ENV", "PFFT_NO_HALF_PAIRS=1").split(","))  # the B leg
    for env in ({}, other):
        p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), str(batch)], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if "N=" in l]
        print(("   %s: " % ",".join("%s=%s" % kv for kv in env.items()) if env else "") + (out[-1] if out else "FAILED: " + p.stderr[-400:]), flush=True)
