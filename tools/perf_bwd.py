"""backward vs forward throughput (same kernels, conjugation on load and store)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import portfft_amd as pf
for name, dims, batch, prec in (("f32 N=4096 x 64Ki", [4096], 65536, "f32"), ("f32 N=1024 x 256Ki", [1024], 262144, "f32"), ("f64 N=4096 x 32Ki", [4096], 32768, "f64"),
                                ("f32 N=1200 x 200Ki (jit)", [1200], 204800, "f32"), ("f32 64x64 x 32Ki (fused)", [64, 64], 32768, "f32"), ("f64 N=2^20 x 128", [1 << 20], 128, "f64")):
    n = 1
    for l in dims: n *= l
    dt = torch.complex64 if prec == "f32" else torch.complex128
    d = pf.descriptor(dims, prec); d.number_of_transforms = batch
    plan = d.commit()
    x = torch.empty(n * batch, dtype=dt, device="cuda"); torch.view_as_real(x).uniform_(-1, 1); y = torch.empty_like(x)
    res = []
    for fn in (plan.compute_forward, plan.compute_backward):
        fn(x, y); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn(x, y)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        res.append(2.0 * n * batch * (8 if prec == "f32" else 16) / ms * 1e-9)
    print("%-30s forward %.2f TB/s   backward %.2f TB/s" % (name, res[0], res[1]))
