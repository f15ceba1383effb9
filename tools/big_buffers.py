"""Buffers beyond 4 GiB / 2^31 elements: every tier must address them with 64-bit offsets.
Device-generated data, sampled batches checked against numpy.  (run on the GPU box)"""
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portfft_amd as pf  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def case(name, n, batch, prec="f32", layout_in="P", layout_out="P", split=False):
    cdt = torch.complex64 if prec == "f32" else torch.complex128
    rdt = torch.float32 if prec == "f32" else torch.float64
    total = n * batch
    g = torch.Generator(device="cuda").manual_seed(1)
    xr = torch.rand(total, 2, dtype=rdt, device="cuda", generator=g) * 2 - 1
    x = torch.view_as_complex(xr)
    d = pf.descriptor([n], prec)
    d.number_of_transforms = batch
    if layout_in == "BI":
        d.forward_strides, d.forward_distance = [batch], 1
    if layout_out == "BI":
        d.backward_strides, d.backward_distance = [batch], 1
    if split:
        d.complex_storage = pf.complex_storage.SPLIT_COMPLEX
    plan = d.commit()
    info = plan.info() if hasattr(plan, "info") else None
    if split:
        a, b = x.real.contiguous(), x.imag.contiguous()
        yr = torch.empty(total, dtype=rdt, device="cuda")
        yi = torch.empty(total, dtype=rdt, device="cuda")
        plan.compute_forward(a, b, yr, yi)
    else:
        y = torch.empty(total, dtype=cdt, device="cuda")
        plan.compute_forward(x, y)
    plan.wait()
    worst = 0.0
    for b_ in sorted({0, 1, batch // 2, batch // 2 + 1, batch - 2, batch - 1, (1 << 32) // (8 * n) + 3 if (1 << 32) // (8 * n) + 3 < batch else 0,
                      (1 << 31) // n + 1 if (1 << 31) // n + 1 < batch else 0}):
        if layout_in == "BI":
            xi_ = x[b_::batch][:n]
        else:
            xi_ = x[b_ * n:(b_ + 1) * n]
        if split:
            if layout_out == "BI":
                yo = torch.complex(yr[b_::batch][:n], yi[b_::batch][:n])
            else:
                yo = torch.complex(yr[b_ * n:(b_ + 1) * n], yi[b_ * n:(b_ + 1) * n])
        else:
            yo = y[b_::batch][:n] if layout_out == "BI" else y[b_ * n:(b_ + 1) * n]
        ref = np.fft.fft(xi_.cpu().numpy().astype(np.complex128))
        worst = max(worst, rel(yo.cpu().numpy().astype(np.complex128), ref))
    gib = total * (8 if prec == "f32" else 16) / 2**30
    ok = worst < (2e-6 if prec == "f32" else 1e-14)
    print(f"{name:42s} n={n:8d} batch={batch:9d} {gib:6.2f} GiB/buffer  worst rel-L2 {worst:.2e}  {'OK' if ok else 'FAIL'}"
          f"  {info if info else ''}", flush=True)
    del x, xr
    torch.cuda.empty_cache()
    return ok


if __name__ == "__main__":
    ok = True
    ok &= case("spec packed 4096", 4096, 300000)
    ok &= case("spec packed 256", 256, 5000000)
    ok &= case("spec packed 16 (register tier)", 16, 80000000)
    ok &= case("spec split 4096", 4096, 300000, split=True)
    ok &= case("BI->BI 4096 (two-stage)", 4096, 300000, layout_in="BI", layout_out="BI")
    ok &= case("BI->P 512 (strided)", 512, 2400000, layout_in="BI")
    ok &= case("P->BI 512 (strided)", 512, 2400000, layout_out="BI")
    ok &= case("generic 1200", 1200, 1000000)
    ok &= case("generic BI 1200", 1200, 1000000, layout_in="BI", layout_out="BI")
    ok &= case("global 2^20 f32", 1 << 20, 1200)
    ok &= case("global 65536 f64", 65536, 10000, prec="f64")
    ok &= case("spec f64 4096", 4096, 150000, prec="f64")
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
