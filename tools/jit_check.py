"""Runtime-specialised kernels: parity vs numpy and throughput for lengths outside the pre-compiled registry."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import portfft_amd as pf

TIER = {0: "REG", 1: "WG", 2: "GENERIC", 3: "GLOBAL"}

def check(n, prec="f32", layout="P", split=False, total=1 << 27, lengths=None):
    cdt = torch.complex64 if prec == "f32" else torch.complex128
    dims = lengths or [n]
    nn = int(np.prod(dims))
    batch = max(16, (total if prec == "f32" else total // 2) // nn)
    batch -= batch % 16
    x = torch.empty(nn * batch, dtype=cdt, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1)
    d = pf.descriptor(dims, prec)
    d.number_of_transforms = batch
    if layout == "BI":
        d.forward_strides, d.forward_distance = [batch], 1
        d.backward_strides, d.backward_distance = [batch], 1
    if split:
        d.complex_storage = pf.complex_storage.SPLIT_COMPLEX
    t0 = time.perf_counter()
    plan = d.commit()
    t_commit = time.perf_counter() - t0
    if split:
        args = [x.real.contiguous(), x.imag.contiguous(), torch.empty(nn * batch, dtype=x.real.dtype, device="cuda"), torch.empty(nn * batch, dtype=x.real.dtype, device="cuda")]
    else:
        args = [x, torch.empty_like(x)]
    plan.compute_forward(*args); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    reps = 5
    for _ in range(reps): plan.compute_forward(*args)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    y = torch.complex(args[2], args[3]) if split else args[1]
    worst = 0.0
    for b in (0, batch // 2 + 1, batch - 1):
        if layout == "BI":
            xi, yo = x[b::batch], y[b::batch]
        else:
            xi, yo = x[b * nn:(b + 1) * nn], y[b * nn:(b + 1) * nn]
        ref = np.fft.fftn(xi.cpu().numpy().astype(np.complex128).reshape(dims)).reshape(-1)
        got = yo.cpu().numpy().astype(np.complex128)
        worst = max(worst, float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
    # backward round trip
    if not split:
        z = torch.empty_like(x)
        plan.compute_backward(y, z); torch.cuda.synchronize()
        rt = float((z[:nn] / nn - x[:nn]).abs().max())
    else:
        rt = 0.0
    info = plan.info()
    tiers = [TIER[info.dims[i].tier] for i in range(len(dims))]
    fac = [list(info.dims[i].factors[:info.dims[i].n_factors]) for i in range(len(dims))]
    tol = 2e-6 if prec == "f32" else 5e-15
    ok = worst < tol and rt < (1e-4 if prec == "f32" else 1e-12)
    esz = 8 if prec == "f32" else 16
    print("%s %-14s %s%s tiers=%s %s wg=%d fpw=%d commit %.2fs  %7.3f ms %5.2f TB/s  rel %.1e rt %.1e %s" % (
        prec, "x".join(map(str, dims)), layout, " split" if split else "", tiers, fac, info.dims[0].workgroup_size,
        info.dims[0].ffts_per_workgroup, t_commit, ms, 2.0 * nn * batch * esz / ms * 1e-9, worst, rt, "OK" if ok else "FAIL"), flush=True)
    return ok

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    ok = True
    if which in ("all", "spec"):
        for n in (6, 7, 12, 24, 30, 31, 48, 60, 120, 243, 343, 625, 720, 1080, 1200, 1920, 2187, 2401, 3000, 3125, 4000, 4800, 6000, 6561, 7429, 7680, 10080, 15625):
            ok &= check(n)
        for n in (6, 24, 31, 48, 120, 243, 625, 720, 1200, 2187, 3000, 3125, 5040, 6561, 7680):
            ok &= check(n, "f64")
    if which in ("all", "other"):
        for n in (120, 625, 1200, 3000):
            ok &= check(n, split=True)
            ok &= check(n, layout="BI")
            ok &= check(n, "f64", layout="BI")
        ok &= check(0, lengths=[600, 1200])
        ok &= check(0, lengths=[30, 50, 70])
        ok &= check(0, "f64", lengths=[243, 625])
        ok &= check(30000); ok &= check(62500); ok &= check(1000000); ok &= check(30000, "f64"); ok &= check(2985984)  # 12^6
        ok &= check(4800, layout="BI"); ok &= check(16000, layout="BI")
    if which in ("all", "small"):
        for n in (4, 8, 16, 31, 32):
            ok &= check(n, layout="BI")
            ok &= check(n, "f64", layout="BI")
        ok &= check(0, lengths=[16, 16, 1024])
        ok &= check(0, lengths=[8, 12, 2048])
        ok &= check(0, "f64", lengths=[16, 16, 512])
        ok &= check(0, lengths=[16, 16, 16, 16])
        ok &= check(0, lengths=[30, 50, 70])
        ok &= check(0, lengths=[12, 64, 64], split=True)
        ok &= check(0, "f64", lengths=[5, 6, 7, 8, 9])
    if which in ("all", "nd"):
        for dims in ([64, 64], [32, 32], [8, 8], [16, 16, 16], [128, 32], [32, 128], [30, 50], [4, 4, 4, 4], [2, 3], [90, 90]):
            ok &= check(0, lengths=dims)
            ok &= check(0, lengths=dims, split=True)
        for dims in ([64, 64], [16, 16, 16], [27, 125], [8, 8]):
            ok &= check(0, "f64", lengths=dims)
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
