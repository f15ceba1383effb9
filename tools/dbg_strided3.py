import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H, gpu_utils as G
import portfft_amd as pf
n, batch = 16384, 128
x, y = H.gen_fourier_data(batch, [n], np.complex128)
d = G.make_descriptor([n], "f64", batch=batch)
plan = d.commit()
for rep in range(3):
    got = G.run(d, pf.direction.FORWARD, x.ravel(), plan=plan).reshape(batch, n)
    errs = np.array([H.rel_l2(got[b], y[b]) for b in range(batch)])
    bad = np.nonzero(errs > 5e-15)[0]
    print("rep", rep, "bad batches", bad[:10], len(bad))
    for b in bad[:3]:
        diff = got[b] - y[b]
        rows = np.nonzero(np.abs(diff.reshape(128, 128)).max(axis=0) > 1e-10)[0]   # k1 = idx % 128
        print("  batch", b, "bad k1 rows", rows)
        for k1 in rows[:2]:
            e = np.fft.ifft(diff[k1::128])      # error of Y[k1][c] (up to the row FFT)
            yrow = np.fft.ifft(y[b][k1::128])
            cols = np.nonzero(np.abs(e) > 1e-12 * np.abs(yrow).max())[0]
            print("    k1", k1, "bad columns c:", cols[:20], len(cols), "rel err of bad entries", (np.abs(e[cols]) / np.abs(yrow[cols]))[:6])
