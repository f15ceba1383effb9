"""Register-resident strided stage kernels (stockham_strided_hx.hpp) through the library: parity against NumPy (double) on the
first and last transforms of the batch and the time per execute (~1 GiB per buffer), for four-step lengths whose stage kernels
sit alone on their CU and for batch-interleaved lengths of the same band.  One process per variant of the environment
(the runtime compiler caches its entries per process): python tools/perf_stage_hx.py [tag] -- the caller sets the knobs
(PFFT_JIT_STRIDED_HX=0 is the LDS-resident twin, PFFT_JIT_STRIDED_HX_FORCE=tpf:per_cu a forced shape, PFFT_XCD_CONTIG=0 ...).
PERF_STAGE_HX_CASES="f32:68640,f32:bi768" selects cases."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import portfft_amd as pf

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
CASES = os.environ.get("PERF_STAGE_HX_CASES",
                       "f32:68640,f32:1000000,f32:62500,f32:2985984,f32:250000,f32:120000,f64:68640,f64:1000000,f64:250000,"
                       "f32:bi660,f32:bi768,f32:bi800,f32:bi1000,f64:bi768,f64:bi660,f32:2d1000").split(",")


def one(case):
    prec, what = case.split(":")
    es = 8 if prec == "f32" else 16
    cdt = torch.complex64 if prec == "f32" else torch.complex128
    if what.startswith("bi"):
        n = int(what[2:].split("@")[0])  # "bi1024@132000": that batch count
        batch = int(what.split("@")[1]) if "@" in what else 4096 * max(1, (1 << 30) // (n * es * 4096))
        lengths = [n]
        d = pf.descriptor(lengths, prec)
        d.number_of_transforms = batch
        d.forward_strides = [batch]
        d.forward_distance = 1
        d.backward_strides = [batch]
        d.backward_distance = 1
        layout = "bi"
    elif what.startswith("2d"):
        n = int(what[2:])
        batch = max(1, (1 << 30) // (n * n * es))
        lengths = [n, n]
        d = pf.descriptor(lengths, prec)
        d.number_of_transforms = batch
        layout = "2d"
    else:
        n = int(what)
        batch = max(1, (1 << 30) // (n * es))
        lengths = [n]
        d = pf.descriptor(lengths, prec)
        d.number_of_transforms = batch
        layout = "p"
    total = batch
    for l in lengths:
        total *= l
    x = torch.empty(total, dtype=cdt, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1)
    y = torch.zeros(total, dtype=cdt, device="cuda")
    plan = d.commit()
    plan.compute_forward(x, y)
    torch.cuda.synchronize()
    # parity: first two and last two transforms against NumPy in double
    per = total // batch
    err = 0.0
    for b in sorted(set([0, 1, batch - 2, batch - 1]) & set(range(batch))):
        if layout == "bi":
            xi = x[b::batch].cpu().numpy().astype(np.complex128)
            yo = y[b::batch].cpu().numpy().astype(np.complex128)
            ref = np.fft.fft(xi)
        elif layout == "2d":
            xi = x[b * per:(b + 1) * per].cpu().numpy().astype(np.complex128).reshape(lengths)
            yo = y[b * per:(b + 1) * per].cpu().numpy().astype(np.complex128).reshape(lengths)
            ref = np.fft.fft2(xi)
        else:
            xi = x[b * per:(b + 1) * per].cpu().numpy().astype(np.complex128)
            yo = y[b * per:(b + 1) * per].cpu().numpy().astype(np.complex128)
            ref = np.fft.fft(xi)
        err = max(err, float(np.linalg.norm(yo - ref) / np.linalg.norm(ref)))
    tol = 2e-6 if prec == "f32" else 5e-15
    reps = 10
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        s.record()
        for _ in range(reps):
            plan.compute_forward(x, y)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    info = plan.info()
    dim = info.dims[len(lengths) - 1]
    frac = 2.0 * total * es / (best * 1e-3) / 8e12
    print("%-10s %-14s frac %.3f  %.4f ms  err %.2e %s  wg %d x%d lds %d factors %s" %
          (tag, case, frac, best, err, "ok" if err < tol else "PARITY-FAIL", dim.workgroup_size, dim.ffts_per_workgroup,
           dim.lds_bytes, [dim.factors[i] for i in range(dim.n_factors)]), flush=True)


for c in CASES:
    try:
        one(c)
    except Exception as ex:  # noqa: BLE001
        print("%-10s %-14s %r" % (tag, c, ex), flush=True)
