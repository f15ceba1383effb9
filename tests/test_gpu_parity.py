"""Parity of the HIP path (through the C ABI) with the oracle, the golden vectors and NumPy.  Needs an MI355X.

Tolerance: relative L2 error per transform <= helpers.REL_L2_TOL (2e-6 fp32, 5e-15 fp64) against the double
precision result -- far inside the 1e-4 bar of BASELINE.json -- plus the reference's own per-element rule
2*eps*N*log2(N) (fft_test_utils.hpp:461-464).  The size / batch / layout grid follows
test/unit_test/instantiate_fft_tests.hpp.
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

F, B = 0, 1


def _pf():
    import portfft_amd as pf
    return pf


def _check(got, ref, n, dtype, what):
    got = np.asarray(got)
    ref = np.asarray(ref)
    tol = H.REL_L2_TOL[np.dtype(dtype)]
    for b in range(got.shape[0]):
        e = H.rel_l2(got[b], ref[b])
        assert e <= tol, (what, "batch", b, "rel-L2", e)
    assert H.check_reference_rule(got.astype(dtype), ref.astype(dtype), n), what


def _golden_cases(golden):
    out = []
    for k in golden["fft"].files:
        if k.endswith("_in"):
            prec, b, dims = k[:-3].split("_")
            out.append((k[:-3], prec, int(b[1:]), [int(x) for x in dims.split("x")]))
    return out


def test_golden_vectors_forward_and_backward(golden, oracle):
    """committed fixtures (reference test generator): GPU == fixture, and GPU == oracle on the same inputs"""
    import gpu_utils as G
    pf = _pf()
    for key, prec, batch, dims in _golden_cases(golden):
        x = golden["fft"][key + "_in"]
        y = golden["fft"][key + "_out"]
        n = int(np.prod(dims))
        d = G.make_descriptor(dims, prec, batch=batch)
        out = G.run(d, pf.direction.FORWARD, x.ravel()).reshape(y.shape)
        _check(out, y, n, x.dtype, key + " fwd vs golden")
        od = oracle.make_desc(dims, prec, batch=batch)
        oref = oracle.compute(od, F, x.ravel(), threads=4).reshape(y.shape)
        _check(out, oref, n, x.dtype, key + " fwd vs oracle")
        back = G.run(d, pf.direction.BACKWARD, y.ravel()).reshape(x.shape)
        _check(back, x.astype(np.complex128) * n, n, x.dtype, key + " bwd vs golden")
    # the reference's GLOBAL-level sizes (GlobalTest / BackwardGlobalTest 32768 / 65536 / 131072, WorkgroupOrGlobal 8192 /
    # 16384 in double: instantiate_fft_tests.hpp:140-151, 169-173) against committed vectors, both placements
    for key, prec, n, x, y in H.golden_global_cases(golden):
        for place in (1, 0):
            d = G.make_descriptor([n], prec, placement=place)
            out = G.run(d, pf.direction.FORWARD, x.ravel()).reshape(y.shape)
            _check(out, y, n, x.dtype, key + " fwd vs golden, placement %d" % place)
            back = G.run(d, pf.direction.BACKWARD, y.ravel()).reshape(x.shape)
            _check(back, x.astype(np.complex128) * n, n, x.dtype, key + " bwd vs golden, placement %d" % place)


def test_config1_in_place(golden):
    """BASELINE config 1: fp32 N=64 batch=1 in-place forward"""
    import gpu_utils as G
    pf = _pf()
    x = golden["fft"]["f32_b1_64_in"]
    y = golden["fft"]["f32_b1_64_out"]
    d = G.make_descriptor([64], "f32", placement=0)
    out = G.run(d, pf.direction.FORWARD, x.ravel())
    assert H.rel_l2(out, y.ravel()) <= 1e-6


SIZES_BATCHES = [
    ([1, 2, 3, 4, 8], [1, 3, 33000]),            # workItemTest
    ([16, 32], [1, 3, 555]),                     # workItemOrSubgroupTest
    ([64, 96, 128], [1, 3, 555]),                # SubgroupTest
    ([256, 512, 1024], [1, 131]),                # SubgroupOrWorkgroupTest
    ([1536], [1, 131]),
    ([2048, 3072, 4096], [1, 3]),                # WorkgroupTest
    ([8192, 16384], [1, 128]),                   # WorkgroupOrGlobal
]
# (placement, input layout, output layout): all_valid_placement_layouts (instantiate_fft_tests.hpp:37-46)
PLACEMENT_LAYOUTS = [(0, "P", "P"), (0, "BI", "BI"), (1, "P", "P"), (1, "P", "BI"), (1, "BI", "BI"), (1, "BI", "P")]


def _layout_desc(G, n, prec, batch, place, lin, lout, direction, storage):
    """descriptor whose input domain has layout lin and output domain lout for `direction`"""
    kw = {}
    fwd_l, bwd_l = (lin, lout) if direction == F else (lout, lin)
    if fwd_l == "BI":
        kw.update(fwd_strides=[batch], fwd_distance=1)
    if bwd_l == "BI":
        kw.update(bwd_strides=[batch], bwd_distance=1)
    return G.make_descriptor([n], prec, batch=batch, storage=storage, placement=place, **kw)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_reference_size_grid_all_layouts(prec, oracle):
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    for sizes, batches in SIZES_BATCHES:
        for n in sizes:
            for batch in batches:
                x, y = H.gen_fourier_data(batch, [n], dtype)
                big = n * batch > 2_000_000
                for place, lin, lout in PLACEMENT_LAYOUTS:
                    if big and (lin, lout) != ("P", "P"):
                        continue
                    if n >= 8192 and (lin, lout) != ("P", "P"):
                        continue  # all_valid_global_placement_layouts: packed only
                    for storage in (0, 1):
                        if storage == 1 and (batch > 200 or n > 4096):
                            continue
                        d = _layout_desc(G, n, prec, batch, place, lin, lout, F, storage)
                        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                        _check(got, y, n, dtype, ("fwd", prec, n, batch, place, lin, lout, storage))
    # SubgroupRegressionTest (instantiate_fft_tests.hpp:114-118): in-place BATCH_INTERLEAVED, interleaved storage,
    # lengths 80 / 100 x batches 44 / 100
    for n in (80, 100):
        for batch in (44, 100):
            x, y = H.gen_fourier_data(batch, [n], dtype)
            d = _layout_desc(G, n, prec, batch, 0, "BI", "BI", F, 0)
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            _check(got, y, n, dtype, ("subgroup regression", prec, n, batch))
    # oracle cross-check on a few mid sizes (seeded inputs, same data on both sides)
    for n, batch in [(64, 3), (1024, 3), (4096, 3), (8192, 1)]:
        x, _ = H.gen_fourier_data(batch, [n], dtype, seed=3)
        d = G.make_descriptor([n], prec, batch=batch)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        ref = oracle.compute(oracle.make_desc([n], prec, batch=batch), F, x.ravel(), threads=4).reshape(x.shape)
        _check(got, ref, n, dtype, ("oracle", prec, n))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_backward_grid(prec):
    """BackwardTest / BackwardGlobalTest (instantiate_fft_tests.hpp:160-173): unnormalised backward"""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    for n in (8, 9, 16, 32, 64, 4096, 32768, 65536):
        for batch in (1, 3):
            x, y = H.gen_fourier_data(batch, [n], dtype)
            layouts = PLACEMENT_LAYOUTS if n <= 4096 else [(0, "P", "P"), (1, "P", "P")]
            for place, lin, lout in layouts:
                for storage in (0, 1):
                    d = _layout_desc(G, n, prec, batch, place, lin, lout, B, storage)
                    got, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                    _check(got, x.astype(np.complex128) * n, n, dtype, ("bwd", prec, n, batch, place, lin, lout))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_global_sizes(prec, oracle):
    """GlobalTest 32768/65536/131072 and the odd composites 9800/15360/68640 (instantiate_fft_tests.hpp:147-157)"""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    for n, batches in [(32768, (1, 3)), (65536, (1, 3)), (131072, (1, 3)), (9800, (3,)), (15360, (3,)), (68640, (3,))]:
        for batch in batches:
            x, y = H.gen_fourier_data(batch, [n], dtype)
            for place in (0, 1):
                d = G.make_descriptor([n], prec, batch=batch, placement=place)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got, y, n, dtype, ("global", prec, n, batch, place))
    x, _ = H.gen_fourier_data(1, [32768], dtype, seed=11)
    got, _ = G.transform_packed(G.make_descriptor([32768], prec), pf.direction.FORWARD, x)
    ref = oracle.compute(oracle.make_desc([32768], prec), F, x.ravel()).reshape(x.shape)
    _check(got, ref, 32768, dtype, "global vs oracle")


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_multidimensional(prec, oracle):
    """MultidimensionalTest (instantiate_fft_tests.hpp:176-182), both directions, both storages"""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    for dims in ([2, 4], [4, 2], [16, 512], [64, 2048], [2, 3, 6], [2, 3, 2, 3]):
        n = int(np.prod(dims))
        for batch in (1, 3):
            x, y = H.gen_fourier_data(batch, dims, dtype)
            for place in (0, 1):
                for storage in (0, 1):
                    d = G.make_descriptor(dims, prec, batch=batch, storage=storage, placement=place)
                    got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                    _check(got.reshape(batch, -1), y.reshape(batch, -1), n, dtype, ("nd fwd", dims, batch, place))
                    back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                    _check(back.reshape(batch, -1), x.reshape(batch, -1).astype(np.complex128) * n, n, dtype,
                           ("nd bwd", dims, batch, place))
    x, _ = H.gen_fourier_data(2, [16, 512], dtype, seed=2)
    got, _ = G.transform_packed(G.make_descriptor([16, 512], prec, batch=2), pf.direction.FORWARD, x)
    ref = oracle.compute(oracle.make_desc([16, 512], prec, batch=2), F, x.ravel(), threads=4).reshape(x.shape)
    _check(got.reshape(2, -1), ref.reshape(2, -1), 16 * 512, dtype, "nd vs oracle")


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_strided_workgroup_tier(prec, oracle):
    """lengths and batch counts that route through the strided work-group kernels (stockham_strided.hpp):
    batch-interleaved on either or both sides, N-D outer dimensions, the two four-step stages of large 1-D
    transforms; ragged groups and offsets included"""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    for n in (64, 128, 256, 512, 1024, 2048):
        for batch in (32, 96, 160):
            x, y = H.gen_fourier_data(batch, [n], dtype)
            for place, lin, lout in PLACEMENT_LAYOUTS[1:]:
                for direction in (F, B):
                    d = _layout_desc(G, n, prec, batch, place, lin, lout, direction, 0)
                    d.forward_offset, d.backward_offset = (5, 5) if place == 0 else (3, 11)
                    src, ref = (x, y) if direction == F else (y, x.astype(np.complex128) * n)
                    got, _ = G.transform_packed(d, pf.direction(direction), src)
                    _check(got, ref, n, dtype, ("strided tier", prec, n, batch, place, lin, lout, direction))
                    if "BI" in (lin, lout):
                        info = d.commit().info()
                        # one strided launch, or (long columns) the two-stage split through scratch
                        assert info.dims[0].tier in (1, 3), (n, batch, lin, lout, info.dims[0].tier)
    # long batch-interleaved transforms: two column-shaped four-step stages through scratch
    for n in (4096, 8192, 16384) if prec == "f32" else (4096, 16384, 65536):
        for batch in (32, 64):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n + batch)
            for place in (0, 1):
                d = _layout_desc(G, n, prec, batch, place, "BI", "BI", F, 0)
                d.forward_scale = 0.5
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got, 0.5 * y, n, dtype, ("BI two-stage fwd", prec, n, batch, place))
                dim = d.commit().info().dims[0]
                # (round 6: a 128-point stage A in front of whatever is left -- the balanced split made stage A of N = 2048 a
                #  single-pass radix-32 kernel with the store modifier, 0.22 of the HBM peak against 0.33)
                assert dim.tier == 3 and dim.factors[0] == 128 and dim.factors[0] * dim.factors[1] == n, (n, list(dim.factors[:2]))
                back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                _check(back, x.astype(np.complex128) * n, n, dtype, ("BI two-stage bwd", prec, n, batch, place))
    for dims in ([256, 256], [64, 1024], [1024, 64], [32, 128, 64]):
        n = int(np.prod(dims))
        x, y = H.gen_fourier_data(2, dims, dtype)
        for place in (0, 1):
            d = G.make_descriptor(dims, prec, batch=2, placement=place, bwd_scale=0.5)
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            _check(got.reshape(2, -1), y.reshape(2, -1), n, dtype, ("nd strided fwd", dims, place))
            back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
            _check(back.reshape(2, -1), x.reshape(2, -1).astype(np.complex128) * n * 0.5, n, dtype, ("nd strided bwd", dims))
    for n in (65536, 1 << 20, 1 << 18):
        x, y = H.gen_fourier_data(2, [n], dtype, seed=4)
        d = G.make_descriptor([n], prec, batch=2, fwd_scale=2.0)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        _check(got, 2.0 * y, n, dtype, ("four-step strided", prec, n))
        back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
        _check(back, x.astype(np.complex128) * n, n, dtype, ("four-step strided bwd", prec, n))
    x, _ = H.gen_fourier_data(1, [65536], dtype, seed=9)
    got, _ = G.transform_packed(G.make_descriptor([65536], prec), pf.direction.FORWARD, x)
    ref = oracle.compute(oracle.make_desc([65536], prec), F, x.ravel()).reshape(x.shape)
    _check(got, ref, 65536, dtype, "four-step vs oracle")


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_long_transforms_in_strided_layouts(prec):
    """Strided / batch-interleaved transforms longer than HALF the LDS (fp32 10 241 ... 20 480 points, fp64 5121 ... 10 240):
    the generic tier needs two images and the four-step plan takes packed data only, so the strided work-group kernel
    runs them one transform per work-group (jit_planner.cpp choose_strided_params, fpw = 1).  Found by tools/fuzz.py seed 61
    (profiles/r5_fuzz_61_150.txt): 11780 / 12464 (fp32) and 5610 (fp64) used to be `unsupported_configuration` inside
    the documented limits."""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    sizes = (11780, 12464, 20480, 16384) if prec == "f32" else (5610, 10240, 7000)
    for n in sizes:
        for batch, lin, lout, place, storage in ((3, "BI", "BI", 1, 0), (33, "P", "BI", 1, 1), (5, "BI", "P", 1, 0),
                                                 (2, "BI", "BI", 0, 0)):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n + batch)
            for direction in (F, B):
                d = _layout_desc(G, n, prec, batch, place, lin, lout, direction, storage)
                src, ref = (x, y) if direction == F else (y, x.astype(np.complex128) * n)
                got, _ = G.transform_packed(d, pf.direction(direction), src)
                _check(got, ref, n, dtype, ("long strided", prec, n, batch, lin, lout, place, storage, direction))
        # a strided row layout (stride 2, padded distance)
        x, y = H.gen_fourier_data(3, [n], dtype, seed=n)
        d = G.make_descriptor([n], prec, batch=3, placement=1, fwd_strides=[2], fwd_distance=2 * n + 5,
                              bwd_strides=[2], bwd_distance=2 * n + 5)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        _check(got, y, n, dtype, ("long strided rows", prec, n))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_register_resident_stage_kernels(prec):
    """The strided tier's register-resident form (stockham_strided_hx.hpp; the reference's one sub-kernel shape for any
    factor of the GLOBAL level, common/global.hpp:135-170, and its BATCH_INTERLEAVED work-group branch,
    workgroup_dispatcher.hpp:148-229): a group that would sit alone on its CU (image > 80 KiB) keeps its values in registers
    and exchanges them through HALF an image, so two work-groups share the CU.  Batch-interleaved lengths of that band
    (ragged passes, partial last groups, both storages, both directions, both placements, offsets and scales) and the
    reference's regression size 68640 (instantiate_fft_tests.hpp:153-157: 104 x 660, stage B on that form) against NumPy and
    against the LDS-resident twin of the same descriptor (PFFT_JIT_STRIDED_HX=0)."""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    tol = 2e-6 if prec == "f32" else 5e-15
    for n in (660, 768):
        plan = _layout_desc(G, n, prec, 48, 1, "BI", "BI", F, 0).commit()
        dim = plan.info().dims[0]
        assert dim.tier == 1 and dim.lds_bytes <= 80 * 1024 and dim.lds_bytes < n * dim.ffts_per_workgroup * (8 if prec == "f32" else 16), \
            ("half image", prec, n, dim.lds_bytes)
        os.environ["PFFT_JIT_STRIDED_HX"] = "0"
        try:
            twin = _layout_desc(G, n, prec, 48, 1, "BI", "BI", F, 0).commit().info().dims[0]
        finally:
            del os.environ["PFFT_JIT_STRIDED_HX"]
        assert twin.lds_bytes > 80 * 1024 and twin.ffts_per_workgroup == dim.ffts_per_workgroup, (prec, n, twin.lds_bytes)
        for batch, lin, lout, place, storage in ((48, "BI", "BI", 1, 0), (133, "BI", "BI", 0, 0), (37, "BI", "BI", 1, 1),
                                                 (64, "P", "BI", 1, 0), (50, "BI", "P", 1, 0)):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n + batch)
            for direction in (F, B):
                d = _layout_desc(G, n, prec, batch, place, lin, lout, direction, storage)
                src, ref = (x, y) if direction == F else (y, x.astype(np.complex128) * n)
                got, _ = G.transform_packed(d, pf.direction(direction), src)
                _check(got, ref, n, dtype, ("stage hx", prec, n, batch, lin, lout, place, storage, direction))
        # offsets and scales through the same kernels
        x, y = H.gen_fourier_data(48, [n], dtype, seed=5)
        d = G.make_descriptor([n], prec, batch=48, placement=1, fwd_strides=[48], fwd_distance=1, bwd_strides=[48],
                              bwd_distance=1, fwd_offset=7, bwd_offset=11, fwd_scale=0.5)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        _check(got, 0.5 * y, n, dtype, ("stage hx offsets + scale", prec, n))
    # four-step: the reference's regression size, stage B (660 points x 16 rows fp32 / 8 rows fp64) on the register-resident form
    n = 68640
    os.environ["PFFT_GLOBAL_N1"] = "104"
    try:
        x, y = H.gen_fourier_data(5, [n], dtype, seed=68640)
        for place in (1, 0):
            d = G.make_descriptor([n], prec, batch=5, placement=place)
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            _check(got, y, n, dtype, ("stage hx four-step fwd", prec, place))
            back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
            _check(back, x.astype(np.complex128) * n, n, dtype, ("stage hx four-step bwd", prec, place))
        info = G.make_descriptor([n], prec, batch=5).commit().info()
        assert list(info.dims[0].factors[:2]) == [104, 660] and info.dims[0].lds_bytes <= 80 * 1024, info.dims[0].lds_bytes
        os.environ["PFFT_JIT_STRIDED_HX"] = "0"
        try:
            twin, _ = G.transform_packed(G.make_descriptor([n], prec, batch=5), pf.direction.FORWARD, x)
        finally:
            del os.environ["PFFT_JIT_STRIDED_HX"]
        assert H.rel_l2(got, twin.astype(np.complex128)) < tol, ("stage hx vs LDS-resident twin", prec)
    finally:
        del os.environ["PFFT_GLOBAL_N1"]


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_wide_register_resident_groups(prec):
    """Batch-interleaved lengths whose full-width group (16 fp32 / 8 fp64 columns) is beyond the LDS but fits the registers of
    ONE work-group (fp32 / fp64 1025 ... 2048 points; the reference's BATCH_INTERLEAVED work-group branch,
    workgroup_dispatcher.hpp:148-229, at lengths it would hand to the GLOBAL level): one HBM pass on the register-resident
    strided kernel with a half image, one work-group per CU -- against NumPy (aligned, unaligned and partial-group batch counts,
    both storages, directions and placements, P <-> BI, offsets + scale, a 2-D array's long columns) and against the
    two-stage twin of the same descriptor (PFFT_NO_BI_WIDE=1)."""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    tol = 2e-6 if prec == "f32" else 5e-15
    es = 8 if prec == "f32" else 16
    full = 16 if prec == "f32" else 8
    # (1536 in fp32: a kernel that spills three registers at its 128 VGPRs, tolerated for these plans; 1728: 36 values per lane)
    for n in (1280, 1536, 1728, 2048):
        dim = _layout_desc(G, n, prec, 48, 1, "BI", "BI", F, 0).commit().info().dims[0]
        assert dim.tier == 1 and dim.ffts_per_workgroup == full and n * full * es > 128 * 1024 and \
            dim.lds_bytes <= 152 * 1024 and dim.n_factors >= 2, ("one pass, full width, half image", prec, n, dim.lds_bytes)
        os.environ["PFFT_NO_BI_WIDE"] = "1"
        try:
            twin_dim = _layout_desc(G, n, prec, 48, 1, "BI", "BI", F, 0).commit().info().dims[0]
        finally:
            del os.environ["PFFT_NO_BI_WIDE"]
        assert twin_dim.tier == 3 or twin_dim.ffts_per_workgroup < full, (prec, n, twin_dim.tier)
        for batch, lin, lout, place, storage in ((48, "BI", "BI", 1, 0), (133, "BI", "BI", 0, 0), (37, "BI", "BI", 1, 1),
                                                 (full, "BI", "BI", 0, 1)):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n + batch)
            for direction in (F, B):
                d = _layout_desc(G, n, prec, batch, place, lin, lout, direction, storage)
                src, ref = (x, y) if direction == F else (y, x.astype(np.complex128) * n)
                got, _ = G.transform_packed(d, pf.direction(direction), src)
                _check(got, ref, n, dtype, ("wide hx", prec, n, batch, lin, lout, place, storage, direction))
        x, y = H.gen_fourier_data(48, [n], dtype, seed=5)
        d = G.make_descriptor([n], prec, batch=48, placement=1, fwd_strides=[48], fwd_distance=1, bwd_strides=[48],
                              bwd_distance=1, fwd_offset=7, bwd_offset=11, fwd_scale=0.5)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        _check(got, 0.5 * y, n, dtype, ("wide hx offsets + scale", prec, n))
        os.environ["PFFT_NO_BI_WIDE"] = "1"
        try:
            twin, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        finally:
            del os.environ["PFFT_NO_BI_WIDE"]
        assert H.rel_l2(got, twin.astype(np.complex128)) < tol, ("wide hx vs two-stage twin", prec, n)
    # SPLIT_COMPLEX at N = 513 ... 1024: the same kernel at DOUBLE width (32 fp32 / 16 fp64 columns: whole 128-byte lines per plane)
    for n in (640, 1024):
        d0 = _layout_desc(G, n, prec, 2 * full + 5, 1, "BI", "BI", F, 1)
        dim = d0.commit().info().dims[0]
        assert dim.tier == 1 and dim.ffts_per_workgroup == 2 * full, ("split storage, double width", prec, n, dim.ffts_per_workgroup)
        os.environ["PFFT_NO_BI_WIDE_SPLIT2"] = "1"
        try:
            assert d0.commit().info().dims[0].ffts_per_workgroup == full, (prec, n)
        finally:
            del os.environ["PFFT_NO_BI_WIDE_SPLIT2"]
        for batch, place in ((2 * full + 5, 1), (133, 0), (2 * full, 1)):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n + batch)
            for direction in (F, B):
                d = _layout_desc(G, n, prec, batch, place, "BI", "BI", direction, 1)
                src, ref = (x, y) if direction == F else (y, x.astype(np.complex128) * n)
                got, _ = G.transform_packed(d, pf.direction(direction), src)
                _check(got, ref, n, dtype, ("wide hx split x2", prec, n, batch, place, direction))
    # the long column dimension of a 2-D array (lengths [2048, 24]: 24 adjacent columns per matrix, 3 matrices)
    x, y = H.gen_fourier_data(3, [2048, 24], dtype, seed=77)
    d = G.make_descriptor([2048, 24], prec, batch=3, placement=1)
    got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
    _check(got, y, 2048 * 24, dtype, ("wide hx 2-D columns", prec))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_every_registered_length(prec):
    """every length that has a specialised kernel (kernels_f32.hip / kernels_f64.hip), plus neighbours that fall to
    the generic tier, packed, ragged batch counts, forward and backward"""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    sizes = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 96, 192, 384, 768, 1536, 3072, 6144,
             12288, 80, 100, 160, 320, 640, 1280, 2560, 5120, 1000, 10000, 6, 12, 15, 17, 19, 23, 29, 31, 49, 121,
             169, 243, 625, 2401, 7 * 11 * 13, 30030 // 2]
    for n in sizes:
        for batch in (1, 5, 67):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n)
            d = G.make_descriptor([n], prec, batch=batch)
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            _check(got, y, n, dtype, ("registered fwd", prec, n, batch))
            back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
            _check(back, x.astype(np.complex128) * n, n, dtype, ("registered bwd", prec, n, batch))


def test_maximum_sizes():
    """the largest single transforms of each tier: 16384 (one work-group, fp32), 2^22 and 2^24 (four-step),
    a 7-smooth length near the top of the generic tier, and a long prime-factor-31 length"""
    import gpu_utils as G
    pf = _pf()
    # (16807 = 7^5, 18000, 19683 = 3^9, 20480, fp64 9604 / 10125 / 10240: the longest single-work-group transforms -- the
    #  whole LDS of a CU; 32768 / fp64 16384: the register-resident kernel's, test_register_resident_lengths)
    for prec, dtype, sizes in (("f32", np.complex64, [1 << 22, 1 << 24, 10080, 31 * 31 * 31 * 8, 9 * 5 * 7 * 11 * 13 * 16,
                                                      16807, 18000, 19683, 20480, 32768]),
                               ("f64", np.complex128, [1 << 22, 5040, 31 * 29 * 23 * 4, 9604, 10125, 10240, 16384])):
        for n in sizes:
            x, y = H.gen_fourier_data(1, [n], dtype, seed=5)
            d = G.make_descriptor([n], prec)
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            _check(got, y, n, dtype, ("max sizes", prec, n))
    with pytest.raises(pf.unsupported_configuration):
        G.make_descriptor([67 * 64]).commit()  # prime factor beyond the wavefront size (test_wave64_prime_factors)
    with pytest.raises(pf.unsupported_configuration):
        G.make_descriptor([1 << 29]).commit()  # beyond the 32-bit byte offsets of one transform (2^28 runs: three stages)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_wave64_prime_factors(prec, oracle):
    """Prime factors 37 ... 61: a wave64 build of the reference (CMakeLists.txt:54 PORTFFT_SUBGROUP_SIZES,
    common/subgroup.hpp:226-253 factorize_sg / fits_in_sg: any factor up to the sub-group size is one cross-lane DFT)
    accepts them; here they are in-register butterflies of the generic tier.  Against NumPy, and against the oracle
    planned with sub-group size 64, forward and backward, both placements, packed and batch-interleaved, a four-step
    length and an N-D shape."""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    for n in (37, 41, 43, 47, 53, 59, 61, 74, 37 * 64, 41 * 64, 61 * 16, 43 * 47, 59 * 59, 3 * 53 * 5, 61 * 61 * 8):
        # (both batches and both placements for the bare primes; the composite lengths -- whose commits dominate the test's
        #  minute -- at batch 5, placement alternating: GPUTEST r05 spent 120 s here)
        for batch in ((1, 5) if n <= 74 else (5,)):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n)
            for place in ((0, 1) if n <= 74 else (n % 2,)):
                d = G.make_descriptor([n], prec, batch=batch, placement=place)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got, y, n, dtype, ("wave64 primes fwd", prec, n, batch, place))
                back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                _check(back, x.astype(np.complex128) * n, n, dtype, ("wave64 primes bwd", prec, n, batch, place))
    for n in (37, 74, 41 * 64, 61 * 16):
        x, _ = H.gen_fourier_data(3, [n], dtype, seed=7)
        got, _ = G.transform_packed(G.make_descriptor([n], prec, batch=3), pf.direction.FORWARD, x)
        ref = oracle.compute(oracle.make_desc([n], prec, batch=3), F, x.ravel(), sg=64).reshape(x.shape)
        _check(got, ref, n, dtype, ("wave64 primes vs oracle(sg=64)", prec, n))
    # the generic tier's "big radix" kernel (what runs when runtime specialisation is off or unavailable)
    os.environ["PFFT_JIT"] = "0"
    try:
        for n in (61, 43 * 47, 53 * 16, 59 * 59 * 8):
            x, y = H.gen_fourier_data(3, [n], dtype, seed=n + 1)
            d = G.make_descriptor([n], prec, batch=3)
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            _check(got, y, n, dtype, ("wave64 primes, generic tier", prec, n))
            back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
            _check(back, x.astype(np.complex128) * n, n, dtype, ("wave64 primes, generic tier bwd", prec, n))
    finally:
        del os.environ["PFFT_JIT"]
    # batch-interleaved and N-D
    x, y = H.gen_fourier_data(33, [37 * 8], dtype, seed=3)
    d = _layout_desc(G, 37 * 8, prec, 33, 1, "BI", "BI", F, 0)
    got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
    _check(got, y, 37 * 8, dtype, ("wave64 primes BI", prec))
    x, y = H.gen_fourier_data(2, [41, 53], dtype, seed=4)
    got, _ = G.transform_packed(G.make_descriptor([41, 53], prec, batch=2), pf.direction.FORWARD, x)
    _check(got.reshape(2, -1), y.reshape(2, -1), 41 * 53, dtype, ("wave64 primes 2-D", prec))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_offsets(prec):
    """Offsets* suites (instantiate_fft_tests.hpp:187-218; instantiated for float and double, :375-403): data starts at
    an offset; everything before the output offset must stay untouched"""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    for (fo, bo) in [(8, 8), (67, 67), (0, 2049), (2049, 0), (2047, 2049)]:
        for direction in (F, B):
            for place, lin, lout in ([(1, "P", "P"), (1, "P", "BI"), (1, "BI", "BI"), (1, "BI", "P")] +
                                     ([(0, "P", "P"), (0, "BI", "BI")] if fo == bo else [])):
                x, y = H.gen_fourier_data(33, [2048], dtype)
                d = _layout_desc(G, 2048, prec, 33, place, lin, lout, direction, 0)
                d.forward_offset, d.backward_offset = fo, bo
                src, ref = (x, y) if direction == F else (y, x.astype(np.complex128) * 2048)
                got, raw = G.transform_packed(d, pf.direction(direction), src)
                _check(got, ref, 2048, dtype, ("offsets", fo, bo, direction, place, lin, lout))
                out_off = bo if direction == F else fo
                if place == 1:
                    assert np.all(raw[:out_off] == H.PADDING_VALUE), "padding before the offset was written"
    # OffsetsWIErrorRegressionTest (:204-209): all_valid_oop_placement_layouts x both_directions x mismatched_offsets at
    # batch 33000 -- the whole 4 x 2 x 3 grid
    x, y = H.gen_fourier_data(33000, [8], dtype)
    for (fo, bo) in [(0, 2049), (2049, 0), (2047, 2049)]:
        for direction in (F, B):
            for lin, lout in (("P", "P"), ("P", "BI"), ("BI", "BI"), ("BI", "P")):
                d = _layout_desc(G, 8, prec, 33000, 1, lin, lout, direction, 0)
                d.forward_offset, d.backward_offset = fo, bo
                src, ref = (x, y) if direction == F else (y, x.astype(np.complex128) * 8)
                got, raw = G.transform_packed(d, pf.direction(direction), src)
                _check(got, ref, 8, dtype, ("offsets wi", fo, bo, direction, lin, lout))
                out_off = bo if direction == F else fo
                assert np.all(raw[:out_off] == H.PADDING_VALUE), "padding before the offset was written"
    # OffsetsMDErrorRegressionTest
    x, y = H.gen_fourier_data(2, [4, 4], dtype)
    d = G.make_descriptor([4, 4], prec, batch=2, fwd_offset=2, bwd_offset=0)
    got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
    _check(got.reshape(2, -1), y.reshape(2, -1), 16, dtype, "offsets md")
    x, y = H.gen_fourier_data(33, [16, 512], dtype)
    for place in (0, 1):
        d = G.make_descriptor([16, 512], prec, batch=33, placement=place, fwd_offset=67, bwd_offset=67)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        _check(got.reshape(33, -1), y.reshape(33, -1), 8192, dtype, "offsets nd")


def test_scales():
    """FwdScaledFFTTest / BwdScaledFFTTest (instantiate_fft_tests.hpp:221-235)"""
    import gpu_utils as G
    pf = _pf()
    for prec, dtype in (("f32", np.complex64), ("f64", np.complex128)):
        for dims in ([9], [16], [64], [512], [4096], [16, 512]):
            n = int(np.prod(dims))
            x, y = H.gen_fourier_data(3, dims, dtype)
            for s in (-1.0, 2.0):
                d = G.make_descriptor(dims, prec, batch=3, fwd_scale=s)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got.reshape(3, -1), (y * s).reshape(3, -1), n, dtype, ("fwd scale", dims, s))
                d = G.make_descriptor(dims, prec, batch=3, bwd_scale=s)
                got, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                _check(got.reshape(3, -1), (x.astype(np.complex128) * n * s).reshape(3, -1), n, dtype,
                       ("bwd scale", dims, s))


def _default_dist(lengths, strides):
    return int(np.prod([l * s for l, s in zip(lengths, strides)]))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_strided_layouts(prec):
    """strided / UNPACKED suites (instantiate_fft_tests.hpp:237-319) at the reference's own batch counts (1, 3, 33000),
    both directions, both storages, float and double (:375-403)"""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    cases = ([(c, (1, 3, 33000), 1) for c in H.STRIDED_OOP_CASES] +
             [(c, (1, 10, 33), 1) for c in H.STRIDED_OOP_BATCH_INTERLEAVED_LIKE] +
             [(c, (1, 3, 33000), 0) for c in H.STRIDED_IP_CASES] +
             [(([3], [66], [66], 2, 2), (1, 3, 33), 0), (([6], [40], [40], 1, 1), (1, 3, 33), 0),
              (([75], [66], [66], 2, 2), (1, 3, 33), 0), (([96], [40], [40], 1, 1), (1, 3, 33), 0),
              # StridedStrideEqualsDistance is all_unpacked_unpacked_layout: both layouts in BOTH placements (:296-301)
              (([8], [2], [2], 2, 2), (1,), 1), (([8], [2], [2], 2, 2), (1,), 0),
              (([8], [1], [1], 1, 1), (1,), 1), (([8], [1], [1], 1, 1), (1,), 0),
              (([4], [4], [4], 3, 3), (4,), 1), (([85], [13], [13], 12, 12), (13,), 0)])
    for (lengths, fs, bs, fd, bd), batches, place in cases:
        n = lengths[0]
        fd = _default_dist(lengths, fs) if fd is None else fd
        bd = _default_dist(lengths, bs) if bd is None else bd
        for batch in batches:
            x, y = H.gen_fourier_data(batch, lengths, dtype)
            for storage in (0, 1):
                d = G.make_descriptor(lengths, prec, batch=batch, storage=storage, placement=place, fwd_strides=fs,
                                      bwd_strides=bs, fwd_distance=fd, bwd_distance=bd)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got, y, n, dtype, ("strided fwd", prec, lengths, fs, bs, fd, bd, batch, storage))
                got, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                _check(got, x.astype(np.complex128) * n, n, dtype, ("strided bwd", lengths, fs, bs, fd, bd, batch))


def test_runtime_specialised_lengths():
    """lengths without a pre-compiled kernel are specialised at commit time (hiprtc, csrc/jit.cpp -- the analogue of
    the reference's specialization constants, committed_descriptor_impl.hpp:448-573): NumPy parity on the packed,
    batch-interleaved, split, N-D and four-step paths, agreement with the runtime-radix generic tier (PFFT_JIT=0),
    and inner counts that are not a multiple of the group width"""
    import os
    import gpu_utils as G
    pf = _pf()
    for prec, dtype, sizes in (("f32", np.complex64, [30, 120, 343, 1200, 3000, 10080]),
                               ("f64", np.complex128, [48, 625, 5040])):
        for n in sizes:
            x, y = H.gen_fourier_data(7, [n], dtype, seed=n)
            d = G.make_descriptor([n], prec, batch=7, bwd_scale=1.0 / n)
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            _check(got, y, n, dtype, ("jit fwd", prec, n))
            back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
            _check(back, x, n, dtype, ("jit bwd", prec, n))
            assert d.commit().info().dims[0].tier in (0, 1), (prec, n)
            os.environ["PFFT_JIT"] = "0"
            try:
                dg = G.make_descriptor([n], prec, batch=7)
                assert dg.commit().info().dims[0].tier == 2, (prec, n)
                generic, _ = G.transform_packed(dg, pf.direction.FORWARD, x)
            finally:
                del os.environ["PFFT_JIT"]
            _check(generic, y, n, dtype, ("generic fwd", prec, n))
            assert H.rel_l2(got, generic) < (2e-6 if prec == "f32" else 5e-15)
    dtype = np.complex64
    for n, batch in ((120, 33), (1200, 20), (3000, 6)):  # 33, 20, 6 columns: partial groups
        x, y = H.gen_fourier_data(batch, [n], dtype, seed=n)
        for storage in (0, 1):
            for lin, lout in (("BI", "BI"), ("BI", "P"), ("P", "BI")):
                d = _layout_desc(G, n, "f32", batch, 1, lin, lout, F, storage)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got, y, n, dtype, ("jit strided", n, batch, lin, lout, storage))
        ds = G.make_descriptor([n], "f32", batch=batch, storage=1)
        got, _ = G.transform_packed(ds, pf.direction.FORWARD, x)
        _check(got, y, n, dtype, ("jit split", n, batch))
    for prec, dtype, dims in (("f32", np.complex64, [60, 150]), ("f32", np.complex64, [6, 10, 14]),
                              ("f64", np.complex128, [27, 125]), ("f32", np.complex64, [48, 1000])):
        n = int(np.prod(dims))
        x, y = H.gen_fourier_data(3, dims, dtype, seed=n)
        d = G.make_descriptor(dims, prec, batch=3)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        _check(got.reshape(3, -1), y.reshape(3, -1), n, dtype, ("jit nd", prec, dims))
        back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
        _check(back.reshape(3, -1), x.reshape(3, -1).astype(np.complex128) * n, n, dtype, ("jit nd bwd", prec, dims))
    for prec, dtype, n in (("f32", np.complex64, 30000), ("f32", np.complex64, 62500), ("f64", np.complex128, 30000),
                           ("f32", np.complex64, 1000000)):
        x, y = H.gen_fourier_data(2, [n], dtype, seed=n)
        d = G.make_descriptor([n], prec, batch=2)
        # (fp32 30000 fits the registers of one work-group: stockham_wg_hx.hpp, one launch, since round 5)
        assert d.commit().info().dims[0].tier == (1 if (prec, n) == ("f32", 30000) else 3)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        _check(got, y, n, dtype, ("jit four-step", prec, n))
        back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
        _check(back, x.astype(np.complex128) * n, n, dtype, ("jit four-step bwd", prec, n))


def test_unpacked_layouts_any_length():
    """UNPACKED layouts (arbitrary stride / distance, instantiate_fft_tests.hpp:237-320) beyond the reference's
    subgroup-sized limit: rows of a padded matrix, every k-th sample, different layouts on the two sides, both
    storages, in place -- on the packed kernels' configurations with runtime strides (stockham_wg_unpacked_kernel)"""
    import gpu_utils as G
    pf = _pf()
    cases = [("f32", 4096, 1, 4160, 1, 4160), ("f32", 4096, 1, 4160, 1, 4096), ("f32", 4096, 2, 8200, 1, 4096),
             ("f32", 1200, 1, 1280, 3, 3700), ("f32", 64, 1, 80, 1, 80), ("f32", 16, 3, 50, 1, 16),
             ("f32", 16, 1, 20, 1, 20),
             ("f64", 4096, 1, 4100, 1, 4100), ("f64", 625, 2, 1300, 1, 640), ("f32", 8192, 1, 8200, 1, 8200),
             ("f64", 4096, 1, 4160, 1, 4096), ("f64", 4096, 2, 8200, 1, 4096), ("f64", 1200, 1, 1280, 3, 3700),
             ("f64", 64, 1, 80, 1, 80), ("f64", 16, 3, 50, 1, 16), ("f64", 16, 1, 20, 1, 20)]
    for prec, n, fs, fd, bs, bd in cases:
        dtype = np.complex64 if prec == "f32" else np.complex128
        for batch in (1, 6, 33):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n + batch)
            for storage in (0, 1):
                d = G.make_descriptor([n], prec, batch=batch, storage=storage, fwd_strides=[fs], fwd_distance=fd,
                                      bwd_strides=[bs], bwd_distance=bd, fwd_offset=5, bwd_offset=2, bwd_scale=1.0 / n)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got, y, n, dtype, ("unpacked fwd", prec, n, fs, fd, bs, bd, batch, storage))
                back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                _check(back, x, n, dtype, ("unpacked bwd", prec, n, fs, fd, bs, bd, batch, storage))
                assert d.commit().info().dims[0].tier in (0, 1)
            if (fs, fd) == (bs, bd):
                d = G.make_descriptor([n], prec, batch=batch, placement=0, fwd_strides=[fs], fwd_distance=fd,
                                      bwd_strides=[bs], bwd_distance=bd)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got, y, n, dtype, ("unpacked in place", prec, n, fs, fd, batch))


def test_fused_multidimensional():
    """N-D transforms that fit LDS run as ONE launch (csrc/stockham_nd.hpp, specialised at commit); the reference
    launches per dimension and per (batch, outer index) (committed_descriptor_impl.hpp:923-948).  Parity with NumPy,
    agreement with the per-dimension path (PFFT_FUSED_ND=0), both storages and placements, offsets, ragged batches"""
    import os
    import gpu_utils as G
    pf = _pf()
    shapes = [("f32", [64, 64]), ("f32", [16, 16, 16]), ("f32", [2, 3]), ("f32", [30, 50]), ("f32", [4, 2, 8]),
              ("f32", [3, 4, 5, 6]), ("f64", [64, 64]), ("f64", [27, 125]), ("f64", [5, 7]),
              # only a suffix of the dimensions fits LDS: fused suffix + strided passes for the rest
              ("f32", [16, 16, 16, 16]), ("f32", [30, 50, 70]), ("f64", [5, 6, 7, 8, 9]), ("f32", [12, 64, 64]),
              # the largest fused shapes (128 KiB of LDS, one work-group per CU)
              ("f32", [128, 128]), ("f64", [64, 128]), ("f32", [8, 128, 128])]
    for prec, dims in shapes:
        dtype = np.complex64 if prec == "f32" else np.complex128
        n = int(np.prod(dims))
        for batch in ((1, 7, 130) if n <= 8192 else (1, 5)):
            x, y = H.gen_fourier_data(batch, dims, dtype, seed=n + batch)
            for storage in (0, 1):
                for place in (0, 1):
                    d = G.make_descriptor(dims, prec, batch=batch, storage=storage, placement=place, fwd_scale=0.25,
                                          fwd_offset=3, bwd_offset=3 if place == 0 else 9)
                    got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                    _check(got.reshape(batch, -1), 0.25 * y.reshape(batch, -1), n, dtype, ("fused nd", prec, dims, batch))
                    back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                    _check(back.reshape(batch, -1), x.reshape(batch, -1).astype(np.complex128) * n, n, dtype,
                           ("fused nd bwd", prec, dims, batch))
        d = G.make_descriptor(dims, prec, batch=7)
        fused, _ = G.transform_packed(d, pf.direction.FORWARD, H.gen_fourier_data(7, dims, dtype, seed=1)[0])
        os.environ["PFFT_FUSED_ND"] = "0"
        try:
            d2 = G.make_descriptor(dims, prec, batch=7)
            per_dim, _ = G.transform_packed(d2, pf.direction.FORWARD, H.gen_fourier_data(7, dims, dtype, seed=1)[0])
        finally:
            del os.environ["PFFT_FUSED_ND"]
        assert H.rel_l2(fused, per_dim) < (2e-6 if prec == "f32" else 5e-15), (prec, dims)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,n", [("f32", 32768), ("f64", 16384), ("f64", 8192), ("f32", 24576), ("f32", 30000), ("f64", 12000),
                                    ("f64", 15000), ("f32", 16384), ("f32", 15360), ("f32", 11264), ("f64", 6144), ("f64", 7680),
                                    # (two-per-CU lengths whose first plan needs scratch at the pair's register budget: the runtime
                                    #  compiler moves on to the next one -- 27.24.18 / 32.28.8, profiles/r5_pair_plans_of_dropped_tuned_entries.txt)
                                    ("f32", 11664), ("f64", 7168), ("f32", 13500)])
def test_register_resident_lengths(prec, n):
    """The 256 KiB transforms that stay in the registers of one work-group for all their passes, one HBM pass
    (stockham_wg_hx.hpp): fp32 32768, the reference's first GlobalTest size, and fp64 16384, its largest
    WorkgroupOrGlobal size (instantiate_fft_tests.hpp:140-151) -- registered kernels -- and lengths whose kernel is planned
    (jit_planner.cpp choose_hx_params: any radices, ragged passes) and compiled at commit.  Against NumPy on every layout the packed kernels
    serve -- both placements, both storages, both directions, offsets and scales, ragged batches -- and against the
    plan the same descriptor gets with PFFT_NO_REGRES=1 (four-step): another algorithm, equal within the tolerance.
    Transforms of 80 ... 152 KiB (fp32 16384 and fp64 8192 registered, the others planned) run the same kernel as TWO
    work-groups per CU; they fit the LDS, so their twin is the LDS-resident kernel of the length."""
    import gpu_utils as G
    pf = _pf()
    dtype = np.complex64 if prec == "f32" else np.complex128
    plan = G.make_descriptor([n], prec, batch=3).commit()
    info = plan.info()
    assert info.dims[0].tier == 1 and list(info.launches) == [1, 1], "one work-group kernel, one launch"
    assert int(np.prod(info.dims[0].factors[:info.dims[0].n_factors])) == n
    os.environ["PFFT_NO_REGRES"] = "1"
    try:
        twin = G.make_descriptor([n], prec, batch=3).commit()
        if n * (8 if prec == "f32" else 16) <= 152 * 1024:  # a pair: it fits the LDS, the twin is the LDS-resident kernel
            assert twin.info().dims[0].tier == 1 and twin.info().dims[0].lds_bytes > 80 * 1024
            assert info.dims[0].lds_bytes <= 80 * 1024 and info.dims[0].workgroup_size <= 512, "two work-groups per CU"
            assert info.dims[0].factors[0] >= 15 and info.dims[0].n_factors == 3
        else:
            assert twin.info().dims[0].tier == 3 and min(twin.info().launches) >= 2
    finally:
        del os.environ["PFFT_NO_REGRES"]
    for batch in (1, 3, 517):  # (517: more transforms than a persistent grid has work-groups, a ragged tail)
        x, y = H.gen_fourier_data(batch, [n], dtype, seed=batch)
        # (every placement x storage at batch 3; the single transform and the long batch on the two combinations that
        #  share nothing: out of place + interleaved, in place + split)
        for place, storage in (((1, 0), (1, 1), (0, 0), (0, 1)) if batch == 3 else ((1, 0), (0, 1))):
            if True:
                d = G.make_descriptor([n], prec, batch=batch, placement=place, storage=storage)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                _check(got, y, n, dtype, ("regres fwd", prec, n, batch, place, storage))
                back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
                _check(back, x.astype(np.complex128) * n, n, dtype, ("regres bwd", prec, n, batch, place, storage))
    x, y = H.gen_fourier_data(5, [n], dtype, seed=77)
    d = G.make_descriptor([n], prec, batch=5, fwd_offset=24, bwd_offset=8, fwd_scale=0.5, bwd_scale=2.0 / n)
    got, buf = G.transform_packed(d, pf.direction.FORWARD, x)
    _check(got, y * 0.5, n, dtype, ("regres offsets + scale", prec, n))
    assert np.all(buf[:8] == H.PADDING_VALUE), "elements in front of the offset stay untouched"
    back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
    _check(back, x.astype(np.complex128) * 2.0, n, dtype, ("regres backward offsets + scale", prec, n))
    # the four-step twin of the same descriptor
    os.environ["PFFT_NO_REGRES"] = "1"
    try:
        got2, _ = G.transform_packed(G.make_descriptor([n], prec, batch=5), pf.direction.FORWARD, x)
    finally:
        del os.environ["PFFT_NO_REGRES"]
    got1, _ = G.transform_packed(G.make_descriptor([n], prec, batch=5), pf.direction.FORWARD, x)
    _check(got1, got2.astype(np.complex128), n, dtype, ("regres vs four-step twin", prec, n))


def test_random_descriptors():
    """seeded random descriptors (rank, 31-smooth lengths, layouts, storages, placements, offsets, scales, precision,
    direction) against NumPy -- the generator of tools/fuzz.py, 16 cases (the GPU suite's time budget; the 200-case run and the
    GLOBAL-tier runs are tools/fuzz.py's own: profiles/r4_fuzz_*.txt)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz.py"), "11", "16"], capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0 and "0 failures" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def test_error_behaviour_and_plan_info():
    import gpu_utils as G
    pf = _pf()
    torch = G.torch_mod()
    d = G.make_descriptor([4096], batch=4)
    plan = d.commit()
    info = plan.info()
    assert info.dims[0].tier == 1 and list(info.dims[0].factors[:3]) == [16, 16, 16]
    x = torch.zeros(4 * 4096, dtype=torch.complex64, device="cuda")
    with pytest.raises(pf.invalid_configuration):  # committed_descriptor_impl.hpp:862-871
        plan.compute_forward(x.real.contiguous(), x.imag.contiguous(), x.real.contiguous(), x.imag.contiguous())
    ds = G.make_descriptor([64], storage=1)
    with pytest.raises(pf.invalid_configuration):
        ds.commit().compute_forward(x, x)
    for case in H.INVALID_CASES[:4]:
        from test_host_api import _pf_desc
        with pytest.raises(pf.invalid_configuration):
            _pf_desc(case).commit()
    with pytest.raises(pf.unsupported_configuration):
        G.make_descriptor([4099]).commit()  # large prime
    assert G.make_descriptor([1 << 20], "f64", batch=2).commit().info().dims[0].tier == 3


def test_streams_and_graph_capture():
    """execute only enqueues kernels on the plan's stream (no allocation, no synchronisation): two plans on two
    streams run concurrently, and an execute can be captured into a HIP graph and replayed on new data"""
    import gpu_utils as G
    pf = _pf()
    torch = G.torch_mod()
    n, batch = 4096, 512
    x, y = H.gen_fourier_data(batch, [n], np.complex64, seed=21)
    x2, y2 = H.gen_fourier_data(batch, [1024, 4], np.complex64, seed=22)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d1 = G.make_descriptor([n], batch=batch)
    d2 = G.make_descriptor([1024, 4], batch=batch)
    p1, p2 = d1.commit(s1), d2.commit(s2)
    a1 = torch.from_numpy(x.ravel()).cuda()
    a2 = torch.from_numpy(x2.ravel()).cuda()
    o1, o2 = torch.empty_like(a1), torch.empty_like(a2)
    torch.cuda.synchronize()
    for _ in range(20):
        p1.compute_forward(a1, o1)
        p2.compute_forward(a2, o2)
    p1.wait()
    p2.wait()
    _check(o1.cpu().numpy().reshape(batch, n), y, n, np.complex64, "stream 1")
    _check(o2.cpu().numpy().reshape(batch, -1), y2.reshape(batch, -1), 4096, np.complex64, "stream 2")
    # graph capture on the plan's stream
    g = torch.cuda.CUDAGraph()
    o1.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        p1.compute_forward(a1, o1)
        p1.compute_backward(o1, a1)   # a1 <- N * a1
    torch.cuda.synchronize()
    a1.copy_(torch.from_numpy(x.ravel()))
    g.replay()
    torch.cuda.synchronize()
    _check(o1.cpu().numpy().reshape(batch, n), y, n, np.complex64, "graph replay 1")
    assert H.rel_l2(a1.cpu().numpy(), x.ravel().astype(np.complex128) * n) < 2e-6
    g.replay()  # second replay consumes the first one's output: forward of (N x)
    torch.cuda.synchronize()
    _check(o1.cpu().numpy().reshape(batch, n), y.astype(np.complex128) * n, n, np.complex64, "graph replay 2")
    # plans whose kernels were compiled at commit time (module launches) capture and replay the same way
    for dims, kw in (([1200], {}), ([64, 64], {}), ([30000], {}), ([4096], dict(fwd_distance=4160, bwd_distance=4160))):
        d = G.make_descriptor(dims, "f32", batch=6, **kw)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            plan = d.commit()
            xin = torch.randn(d.get_input_count(pf.direction.FORWARD), dtype=torch.complex64, device="cuda")
            out = torch.zeros(d.get_output_count(pf.direction.FORWARD), dtype=torch.complex64, device="cuda")
            plan.compute_forward(xin, out)
            s.synchronize()
            ref = out.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                plan.compute_forward(xin, out)
            out.zero_()
            g.replay()
            s.synchronize()
        assert torch.equal(ref, out), dims


def test_full_size_config2_properties():
    """BASELINE config 2 at full size (fp32 N=4096 batch=65536, 2 GiB in / 2 GiB out): sampled batches against
    NumPy, Parseval on every batch, forward->backward round trip, linearity."""
    import gpu_utils as G
    pf = _pf()
    torch = G.torch_mod()
    n, batch = 4096, 65536
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1, generator=g)
    y = torch.empty_like(x)
    d = G.make_descriptor([n], batch=batch)
    plan = d.commit()
    plan.compute_forward(x, y).wait()
    xs, ys = x.view(batch, n), y.view(batch, n)
    for b in [0, 1, 4095, 32768, 65535] + list(np.random.default_rng(0).integers(0, batch, 27)):
        ref = np.fft.fft(xs[b].cpu().numpy().astype(np.complex128))
        assert H.rel_l2(ys[b].cpu().numpy(), ref) <= 2e-6, b
    # Parseval, every batch
    ex = (xs.abs().double() ** 2).sum(dim=1)
    ey = (ys.abs().double() ** 2).sum(dim=1)
    assert float(((ey / (n * ex)) - 1).abs().max()) < 1e-5
    # round trip: backward(forward(x)) == N x
    z = torch.empty_like(x)
    plan.compute_backward(y, z).wait()
    err = (z.view(batch, n) / n - xs).abs().double().pow(2).sum(dim=1).sqrt() / ex.sqrt()
    assert float(err.max()) < 2e-6
    # linearity: F(2x + w) == 2 F(x) + F(w)
    w = torch.empty_like(x)
    torch.view_as_real(w).uniform_(-1, 1, generator=g)
    plan.compute_forward(w, z).wait()          # z = F(w)
    w.add_(x, alpha=2.0)                       # w = 2x + w
    x2 = torch.empty_like(x)
    plan.compute_forward(w, x2).wait()         # x2 = F(2x + w)
    lin = (x2 - 2 * y - z).view(batch, n).abs().double().pow(2).sum(dim=1).sqrt()
    nrm = x2.view(batch, n).abs().double().pow(2).sum(dim=1).sqrt()
    assert float((lin / nrm).max()) < 2e-6
