import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H, gpu_utils as G
import portfft_amd as pf
for n, batch, place in [(16384, 128, 1), (16384, 128, 0), (16384, 40, 1), (16384, 1, 1), (65536, 8, 1)]:
    x, y = H.gen_fourier_data(batch, [n], np.complex128)
    d = G.make_descriptor([n], "f64", batch=batch, placement=place)
    got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
    errs = np.array([H.rel_l2(got[b], y[b]) for b in range(batch)])
    bad = np.nonzero(errs > 5e-15)[0]
    print(n, batch, place, "max err %.3e" % errs.max(), "bad batches:", bad[:20], len(bad))
    if len(bad):
        b = bad[0]; diff = np.abs(got[b] - y[b]); idx = np.nonzero(diff > 1e-9 * np.abs(y[b]).max())[0]
        print("   bad element indices:", idx[:20], len(idx), " k1 = idx %% n1:", (idx % 128)[:10], " k2:", (idx // 128)[:10])
