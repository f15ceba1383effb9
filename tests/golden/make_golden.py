#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ (run in the build container only).

1. fft_vectors.npz -- input/output pairs produced exactly the way the reference's own tests produce their
   expected values (test/common/reference_data_wrangler.hpp:117-145): rng = Generator(SFC64(0)), real block drawn
   before the imaginary block from uniform(-1, 1), cast to the test precision, np.fft.fftn over the FFT axes in
   double, cast back.  The size/batch grid follows test/unit_test/instantiate_fft_tests.hpp:95-182 (batch sizes
   trimmed so the fixture stays small).
2. static_twiddles.npz -- the reference's twiddle<T>::Re/Im table, obtained by IMPORTING the reference's generator
   /root/reference/scripts/generate_twiddles.py (generate(64), :60-92).  Only its numeric output is stored.

The reference tree is not available on the GPU box, so tests read these files, never /root/reference.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"


def gen_data(batch, dims, is_double):
    """Same statements as the inline script of reference_data_wrangler.hpp:120-145."""
    scalar_type = np.double if is_double else np.single
    complex_type = np.complex128 if is_double else np.complex64
    data_gen_dims = [batch] + list(dims)
    rng = np.random.Generator(np.random.SFC64(0))
    in_data = rng.uniform(-1, 1, data_gen_dims).astype(scalar_type)
    in_data = in_data + 1j * rng.uniform(-1, 1, data_gen_dims).astype(scalar_type)
    out_data = np.fft.fftn(in_data, axes=range(1, len(dims) + 1))
    out_data = out_data.astype(complex_type)
    return in_data.astype(complex_type), out_data


# (batch, dims) -- see instantiate_fft_tests.hpp: workItemTest 1,2,3,4,8; 16,32; SubgroupTest 64,96,128;
# regression 80,100; 256,512,1024; 1536; WorkgroupTest 2048,3072,4096; 8192,16384; BackwardTest 8,9,...;
# MultidimensionalTest {2,4},{4,2},{16,512},{2,3,6},{2,3,2,3}; plus BASELINE config C1 (N=64, batch 1).
CASES = [(1, [64])]
CASES += [(3, [n]) for n in (1, 2, 3, 4, 5, 7, 8, 9, 16, 27, 32, 56, 64, 80, 96, 100, 128, 256, 512, 1024, 1536)]
CASES += [(3, [n]) for n in (2048, 3072, 4096)]
CASES += [(1, [n]) for n in (8192, 9800, 15360, 16384)]
CASES += [(3, [2, 4]), (3, [4, 2]), (1, [16, 512]), (3, [2, 3, 6]), (3, [2, 3, 2, 3])]


def key(prec, batch, dims):
    return "%s_b%d_%s" % (prec, batch, "x".join(str(d) for d in dims))


def main():
    vectors = {}
    for is_double in (False, True):
        prec = "f64" if is_double else "f32"
        for batch, dims in CASES:
            if is_double and int(np.prod(dims)) > 4096:
                continue
            inp, out = gen_data(batch, dims, is_double)
            vectors[key(prec, batch, dims) + "_in"] = inp
            vectors[key(prec, batch, dims) + "_out"] = out
    np.savez_compressed(os.path.join(HERE, "fft_vectors.npz"), **vectors)

    spec = importlib.util.spec_from_file_location("generate_twiddles",
                                                  os.path.join(REFERENCE, "scripts", "generate_twiddles.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    real, imag = mod.generate(64)
    np.savez_compressed(os.path.join(HERE, "static_twiddles.npz"), re=np.array(real, dtype=np.float64),
                        im=np.array(imag, dtype=np.float64))
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in ("fft_vectors.npz", "static_twiddles.npz"))
    print("wrote %d arrays, %d bytes" % (len(vectors), total))


if __name__ == "__main__":
    sys.exit(main())
