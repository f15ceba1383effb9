"""accuracy (rel-L2 and worst element) of the four-step tier and long batch-interleaved transforms vs NumPy in double"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import portfft_amd as pf
from perf_survey_lib import run
def acc(name, n, prec="f32", batch=4, **kw):
    cdt = torch.complex64 if prec == "f32" else torch.complex128
    d = pf.descriptor([n], prec); d.number_of_transforms = batch
    for k, v in kw.items(): setattr(d, k, v)
    x = torch.empty(n * batch, dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(-1, 1)
    y = torch.empty_like(x)
    d.commit().compute_forward(x, y).wait()
    bi = "forward_strides" in kw
    xs = x.cpu().numpy().astype(np.complex128); ys = y.cpu().numpy().astype(np.complex128)
    xs = xs.reshape(n, batch).T if bi else xs.reshape(batch, n)
    ys = ys.reshape(n, batch).T if bi else ys.reshape(batch, n)
    ref = np.fft.fft(xs, axis=1)
    rel = np.linalg.norm(ys - ref) / np.linalg.norm(ref)
    worst = np.max(np.abs(ys - ref)) / np.sqrt(np.mean(np.abs(ref) ** 2))
    print("%-28s rel-L2 %.2e   worst element / rms %.2e" % (name, rel, worst), flush=True)
acc("f32 N=65536", 65536); acc("f32 N=2^20", 1 << 20); acc("f32 N=2^22", 1 << 22); acc("f32 N=30000", 30000); acc("f32 N=10^6", 1000000)
acc("f32 BI N=1200", 1200, batch=64, forward_strides=[64], forward_distance=1, backward_strides=[64], backward_distance=1)
acc("f32 BI N=4096", 4096, batch=64, forward_strides=[64], forward_distance=1, backward_strides=[64], backward_distance=1)
acc("f64 N=2^20", 1 << 20, "f64"); acc("f64 N=65536", 65536, "f64")
run("f64 N=2^20 b=128 (C3)", [1 << 20], 128, "f64", reps=10)
run("f32 N=65536 b=2Ki", [65536], 2 << 10, reps=10)
run("f32 N=2^20 b=256", [1 << 20], 256, reps=10)
