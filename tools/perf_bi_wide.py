"""BATCH_INTERLEAVED lengths beyond the LDS at full group width (fp32 1025 ... 2048 points x 16 columns, fp64 x 8): the one-pass
plan on the one-per-CU register-resident strided kernel (round 6, plan_global.cpp / jit_strided_kernel with a negative group
width) against its two-stage twin (PFFT_NO_BI_WIDE=1).  Checks the first and the last transforms against NumPy in double, then
times BI -> BI.  usage: python tools/perf_bi_wide.py [f32|f64|both] [n ...]
PFFT_PERF_SPLIT=1: SPLIT_COMPLEX storage; PFFT_PERF_QUICK=1: the aligned ~1 GiB batch count only; PFFT_PERF_BATCH=<b>: that batch
count (arrays of 4 GiB and more: the BIG forms)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import portfft_amd as pf

precs = ("f32", "f64")
args = sys.argv[1:]
if args and args[0] in ("f32", "f64", "both"):
    precs = ("f32", "f64") if args[0] == "both" else (args[0],)
    args = args[1:]
lengths = [int(a) for a in args] or [1280, 1536, 2000, 2048]
mode = "two-stage" if os.environ.get("PFFT_NO_BI_WIDE") == "1" else "wide"
split = os.environ.get("PFFT_PERF_SPLIT") == "1"
quick = os.environ.get("PFFT_PERF_QUICK") == "1"
for prec in precs:
    esz = 8 if prec == "f32" else 16
    dt = torch.complex64 if prec == "f32" else torch.complex128
    for n in lengths:
        b1 = (1 << 30) // (n * esz) // 64 * 64
        if os.environ.get("PFFT_PERF_BATCH"):
            batches = [int(os.environ["PFFT_PERF_BATCH"])]
        else:
            batches = [b1] if quick else sorted({b1, b1 + 5, 33000})
        for b in batches:
            d = pf.descriptor([n], prec)
            d.number_of_transforms = b
            d.forward_strides = [b]; d.forward_distance = 1
            d.backward_strides = [b]; d.backward_distance = 1
            if split:
                d.complex_storage = pf.complex_storage.SPLIT_COMPLEX
            x = torch.empty(n * b, dtype=dt, device="cuda"); torch.view_as_real(x).uniform_(-1, 1)
            y = torch.empty_like(x)
            plan = d.commit()
            if split:
                xr, xi = x.real.contiguous(), x.imag.contiguous()
                yr, yi = torch.empty_like(xr), torch.empty_like(xi)
                fwd = lambda: plan.compute_forward(xr, xi, yr, yi)
                fwd(); torch.cuda.synchronize()
                y = torch.complex(yr, yi)
            else:
                fwd = lambda: plan.compute_forward(x, y)
                fwd(); torch.cuda.synchronize()
            xs = x.view(n, b); ys = y.view(n, b)
            cols = list(range(0, 20)) + list(range(b - 20, b))
            ref = np.fft.fft(xs[:, cols].cpu().numpy().astype(np.complex128), axis=0)
            got = ys[:, cols].cpu().numpy()
            err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
            if split:
                br, bi = torch.empty_like(xr), torch.empty_like(xi)
                plan.compute_backward(yr, yi, br, bi); torch.cuda.synchronize()
                back = torch.complex(br, bi)
            else:
                back = torch.empty_like(x)
                plan.compute_backward(y, back); torch.cuda.synchronize()
            rt = (torch.linalg.norm(back / n - x) / torch.linalg.norm(x)).item()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 8 if n * b * esz < (3 << 30) else 3
            s.record()
            for _ in range(reps): fwd()
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / reps
            info = plan.info()
            di = info.dims[0]
            print("%-9s %s%s N=%-5d batch %-7d tier=%d factors=%s wg=%d fpw=%d  %8.4f ms  frac %.3f  err %.2e  round-trip %.2e" % (
                mode, prec, " split" if split else "", n, b, di.tier, list(di.factors[:di.n_factors]), di.workgroup_size, di.ffts_per_workgroup, ms,
                2.0 * n * b * esz / ms * 1e-9 / 8, err, rt), flush=True)
            del x, y, back, plan
