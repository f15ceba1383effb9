import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H, gpu_utils as G
import portfft_amd as pf
for prec, dtype in (("f64", np.complex128), ("f32", np.complex64)):
  for n, batch in [(128, 16384), (128, 4096), (256, 8192), (1024, 2048)]:
    x, y = H.gen_fourier_data(batch, [n], dtype)
    for name, kw in (("BI->BI", dict(fwd_strides=[batch], fwd_distance=1, bwd_strides=[batch], bwd_distance=1)),
                     ("P->BI", dict(bwd_strides=[batch], bwd_distance=1)), ("BI->P", dict(fwd_strides=[batch], fwd_distance=1))):
        d = G.make_descriptor([n], prec, batch=batch, **kw)
        for rep in range(2):
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            errs = np.array([H.rel_l2(got[b], y[b]) for b in range(batch)])
            tol = 5e-15 if prec == "f64" else 2e-6
            bad = np.nonzero(errs > tol)[0]
            print(prec, n, batch, name, "tier", d.commit().info().dims[0].tier, "max err %.3e" % errs.max(), "bad:", bad[:12], len(bad))
