"""three-stage plan with offsets, scales, both directions and placements against torch.fft (complex128)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import portfft_amd as pf
n, batch = 1 << 23, 2
for prec, ct in (("f32", torch.complex64), ("f64", torch.complex128)):
    for place in (pf.placement.OUT_OF_PLACE, pf.placement.IN_PLACE):
        d = pf.descriptor([n], prec)
        d.number_of_transforms = batch
        d.placement = place
        d.forward_offset = 7
        d.backward_offset = 7 if place == pf.placement.IN_PLACE else 3
        d.forward_scale = 0.5
        d.backward_scale = 0.25
        plan = d.commit()
        assert plan.info().dims[0].n_factors == 3
        x = torch.empty(batch * n + 16, dtype=ct, device="cuda"); torch.view_as_real(x).uniform_(-1, 1)
        for direction in ("fwd", "bwd"):
            io, oo = (7, d.backward_offset) if direction == "fwd" else (d.backward_offset, 7)
            src = x[io:io + batch * n].view(batch, n).to(torch.complex128)
            ref = (torch.fft.fft(src, dim=1) * 0.5) if direction == "fwd" else (torch.fft.ifft(src, dim=1) * n * 0.25)
            if place == pf.placement.IN_PLACE:
                w = x.clone()
                (plan.compute_forward if direction == "fwd" else plan.compute_backward)(w)
                got = w
            else:
                got = torch.full_like(x, 9.0)
                (plan.compute_forward if direction == "fwd" else plan.compute_backward)(x, got)
            torch.cuda.synchronize()
            g = got[oo:oo + batch * n].view(batch, n).to(torch.complex128)
            err = ((g - ref).norm() / ref.norm()).item()
            untouched = True if place == pf.placement.IN_PLACE else bool((got[:oo] == 9.0).all() and (got[oo + batch * n:] == 9.0).all())
            print(prec, "in-place" if place == pf.placement.IN_PLACE else "out-of-place", direction, "err %.2e" % err, "guard ok" if untouched else "GUARD OVERWRITTEN")
            assert err < (2e-6 if prec == "f32" else 1e-14) and untouched
print("three-stage offsets/scales OK")
