"""SPLIT_COMPLEX four-step plans: group-major intermediate + tiled-input mixed stage B against the round's earlier plan
(PFFT_NO_SPLIT_TILED=1) -- error against torch.fft and time, in child processes"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, os, torch
sys.path.insert(0, os.path.join(%r, "..", ".."))
import portfft_amd as pf
prec, n, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rt = torch.float32 if prec == "f32" else torch.float64
d = pf.descriptor([n], prec); d.number_of_transforms = batch; d.placement = pf.placement.OUT_OF_PLACE
d.complex_storage = pf.complex_storage.SPLIT_COMPLEX
plan = d.commit()
re = torch.empty(batch * n, dtype=rt, device="cuda").uniform_(-1, 1); im = torch.empty(batch * n, dtype=rt, device="cuda").uniform_(-1, 1)
ore, oim = torch.empty_like(re), torch.empty_like(im)
plan.compute_forward(re, im, ore, oim); torch.cuda.synchronize()
nb = min(batch, 3)
x = torch.complex(re.view(batch, n)[:nb].double(), im.view(batch, n)[:nb].double())
ref = torch.fft.fft(x, dim=1)
y = torch.complex(ore.view(batch, n)[:nb].double(), oim.view(batch, n)[:nb].double())
err = ((y - ref).norm() / ref.norm()).item()
zre, zim = torch.empty_like(re), torch.empty_like(im)
plan.compute_backward(ore, oim, zre, zim); torch.cuda.synchronize()
z = torch.complex(zre.view(batch, n)[:nb].double(), zim.view(batch, n)[:nb].double()) / n
errb = ((z - x).norm() / x.norm()).item()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): plan.compute_forward(re, im, ore, oim)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("%%s split N=%%d b=%%d: %%.4f ms  %%.3f of peak  err %%.2e  roundtrip %%.2e" %% (prec, n, batch, ms, 4 * re.numel() * re.element_size() / (ms * 1e-3) / 8e12, err, errb))
''' % HERE
cases = [("f32", 1 << 16), ("f32", 1 << 18), ("f32", 1 << 20), ("f32", 1 << 17), ("f32", 1 << 19), ("f32", 3 << 18), ("f32", 1000000), ("f32", 640000),
         ("f64", 1 << 16), ("f64", 1 << 18), ("f64", 1 << 20), ("f64", 3 << 16)]
if len(sys.argv) > 1:
    cases = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
for prec, n in cases:
    batch = max(1, ((128 << 20) if prec == "f32" else (64 << 20)) // n)
    for env in ({}, dict(kv.split("=") for kv in os.environ.get("AB_ENV", "PFFT_NO_SPLIT_TILED=1").split(","))):
        p = subprocess.run([sys.executable, "-c", CHILD, prec, str(n), str(batch)], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if "N=" in l]
        print(("   %s: " % ",".join("%s=%s" % kv for kv in env.items()) if env else "") + (out[-1] if out else "FAILED: " + p.stderr[-600:]), flush=True)
