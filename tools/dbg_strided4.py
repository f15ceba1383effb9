import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H, gpu_utils as G
import portfft_amd as pf
for prec, dtype, tol in (("f64", np.complex128, 5e-15), ("f32", np.complex64, 4e-7)):
  for n, batch in [(65536, 64), (16384, 256), (16384, 31), (1 << 20, 4), (4096 * 4, 64)]:
    x, y = H.gen_fourier_data(batch, [n], dtype)
    d = G.make_descriptor([n], prec, batch=batch)
    plan = d.commit()
    inf = plan.info().dims[0]
    got = G.run(d, pf.direction.FORWARD, x.ravel(), plan=plan).reshape(batch, n)
    errs = np.array([H.rel_l2(got[b], y[b]) for b in range(batch)])
    bad = np.nonzero(errs > tol)[0]
    print(prec, n, batch, "factors", list(inf.factors[:2]), "fpw", inf.ffts_per_workgroup, "max err %.3e median %.3e" % (errs.max(), np.median(errs)), "bad:", bad[:10], len(bad))
