"""Shared test helpers: reference-style data generation, layout scatter/gather, error metrics, test tables."""
import numpy as np

FORWARD, BACKWARD = 0, 1
PADDING_VALUE = -5.0  # test/unit_test/fft_test_utils.hpp:452


def gen_fourier_data(batch, dims, dtype, seed=0):
    """The reference test generator (test/common/reference_data_wrangler.hpp:117-145): (forward data, backward data)
    as packed arrays of shape [batch, *dims]."""
    is_double = np.dtype(dtype) in (np.dtype(np.complex128), np.dtype(np.float64))
    scalar = np.double if is_double else np.single
    ctype = np.complex128 if is_double else np.complex64
    shape = [batch] + list(dims)
    rng = np.random.Generator(np.random.SFC64(seed))
    x = rng.uniform(-1, 1, shape).astype(scalar)
    x = x + 1j * rng.uniform(-1, 1, shape).astype(scalar)
    y = np.fft.fftn(x, axes=range(1, len(dims) + 1))
    return x.astype(ctype), y.astype(ctype)


def golden_global_cases(golden):
    """(key, precision, n, input, expected output) of the GLOBAL-level fixture (tests/golden/fft_vectors_global.npz:
    the reference's GlobalTest / WorkgroupOrGlobal sizes, batch 1).  Only the output is stored; the input is regenerated
    from the reference generator's seed and must hash to the stored sha256."""
    import hashlib
    g = golden["global"]
    for k in sorted(f[:-4] for f in g.files if f.endswith("_out")):
        prec, _, n = k.split("_")
        x, _ = gen_fourier_data(1, [int(n)], np.complex64 if prec == "f32" else np.complex128)
        digest = hashlib.sha256(np.ascontiguousarray(x).tobytes()).digest()
        assert digest == g[k + "_in_sha256"].tobytes(), "the regenerated input of %s is not the fixture's" % k
        yield k, prec, int(n), x, g[k + "_out"]


def default_strides(dims):
    s, t = [0] * len(dims), 1
    for i in reversed(range(len(dims))):
        s[i] = t
        t *= dims[i]
    return s


def element_indices(batch, dims, strides, distance, offset):
    """flat index of every element of a [batch, *dims] array in a strided buffer"""
    idx = offset + np.arange(batch, dtype=np.int64).reshape([batch] + [1] * len(dims)) * distance
    for ax, (n, s) in enumerate(zip(dims, strides)):
        shape = [1] * (len(dims) + 1)
        shape[ax + 1] = n
        idx = idx + np.arange(n, dtype=np.int64).reshape(shape) * s
    return idx


def scatter(packed, strides, distance, offset, count, pad=PADDING_VALUE):
    """lay a packed [batch, *dims] array out in a flat buffer of `count` elements (padding value elsewhere), as
    reference_data_wrangler.hpp:52-90 does"""
    buf = np.full(count, pad, dtype=packed.dtype)
    idx = element_indices(packed.shape[0], packed.shape[1:], strides, distance, offset)
    buf[idx.ravel()] = packed.ravel()
    return buf


def gather(buf, batch, dims, strides, distance, offset):
    idx = element_indices(batch, dims, strides, distance, offset)
    return buf[idx.ravel()].reshape([batch] + list(dims))


def rel_l2(a, b):
    a = np.asarray(a).astype(np.complex128).ravel()
    b = np.asarray(b).astype(np.complex128).ravel()
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))


def reference_tolerance(dtype, n):
    """per-element abs-or-rel tolerance of the reference tests: 2 * eps * N * log2(N) (fft_test_utils.hpp:461-464)"""
    eps = np.finfo(np.float64 if np.dtype(dtype) == np.dtype(np.complex128) else np.float32).eps
    return 2.0 * eps * n * max(np.log2(n), 1.0)


def check_reference_rule(out, ref, n):
    """reference_data_wrangler.hpp:355-370: every element within tol absolutely OR relatively"""
    tol = reference_tolerance(np.asarray(ref).dtype if np.asarray(ref).dtype.kind == "c" else np.complex64, n)
    out = np.asarray(out).astype(np.complex128).ravel()
    ref = np.asarray(ref).astype(np.complex128).ravel()
    absd = np.abs(out - ref)
    reld = absd / np.maximum(np.abs(ref), 1e-300)
    return bool(np.all((absd <= tol) | (reld <= tol)))


# FP tolerance of this repo's parity tests: relative L2 error per transform against the double-precision result.
# north_star asks for <= 1e-4; we hold the implementation to the precision it actually has.
REL_L2_TOL = {np.dtype(np.complex64): 2e-6, np.dtype(np.complex128): 5e-15}


# invalid descriptors of test/unit_test/instantiate_fft_tests.hpp:322-373:
# (lengths, fwd_strides, bwd_strides, fwd_distance, bwd_distance, batch, placement) -- None = default
IN_PLACE, OUT_OF_PLACE = 0, 1
INVALID_CASES = [
    ("InvalidLength", [0], None, None, None, None, 1, OUT_OF_PLACE),
    ("InvalidBatch", [1], None, None, None, None, 0, OUT_OF_PLACE),
    ("InvalidDistance0", [5], [5], [1], 0, 5, 2, OUT_OF_PLACE),
    ("InvalidDistance1", [5], [1], [5], 5, 0, 2, OUT_OF_PLACE),
    ("InvalidNonPositiveStrides0", [5], [0], [1], None, None, 1, OUT_OF_PLACE),
    ("InvalidNonPositiveStrides1", [5], [1], [0], None, None, 1, OUT_OF_PLACE),
    ("InvalidNonPositiveStrides2", [5, 12], [12, 1], [12, 0], None, None, 1, OUT_OF_PLACE),
    ("InvalidShortDistance0", [8], [1], [1], 7, 8, 2, OUT_OF_PLACE),
    ("InvalidShortDistance1", [8, 4], [8, 2], [4, 1], 24, 24, 2, OUT_OF_PLACE),
    ("InvalidIPNotMatching0", [8], [2], [1], 16, 8, 2, IN_PLACE),
    ("InvalidIPNotMatching1", [8, 4], [8, 2], [8, 2], 48, 50, 2, IN_PLACE),
    ("InvalidOverlap0", [4], [1], [1], 1, 4, 3, OUT_OF_PLACE),
    ("InvalidOverlap1", [4], [1], [2], 4, 3, 3, OUT_OF_PLACE),
    ("InvalidOverlapLarge", [8], [3333333], [3333333], 1, 1, 3333334, OUT_OF_PLACE),
    ("InvalidStrideEqualsDistance0", [8], [2], [2], 2, 2, 2, OUT_OF_PLACE),
    ("InvalidStrideEqualsDistance1", [8], [1], [1], 1, 1, 2, OUT_OF_PLACE),
]

# valid strided layouts of instantiate_fft_tests.hpp:237-319: (lengths, fwd_strides, bwd_strides, fwd_dist, bwd_dist)
STRIDED_OOP_CASES = [
    ([3], [4], [7], None, None),
    ([8], [11], [2], None, None),
    ([9], [3], [4], 30, 40),
    ([64], [1], [7], None, None),
    ([64], [4], [7], None, None),
    ([75], [3], [2], 300, 200),
    ([104], [3], [4], None, None),
]
STRIDED_OOP_BATCH_INTERLEAVED_LIKE = [
    ([8], [33], [99], 1, 3),
    ([8], [33], [2], 1, 16),
    ([8], [2], [66], 16, 2),
    ([64], [33], [99], 1, 3),
    ([96], [33], [2], 1, 192),
    ([70], [2], [66], 140, 2),
]
STRIDED_IP_CASES = [
    ([3], [4], [4], None, None),
    ([9], [3], [3], 25, 25),
    ([75], [4], [4], None, None),
    ([96], [3], [3], 286, 286),
]
