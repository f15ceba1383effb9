"""Pins the CPU oracle (oracle/) against the reference's own golden data.  CPU only.

  * NumPy vectors generated the way the reference tests generate theirs (tests/golden/fft_vectors.npz),
  * the static twiddle table obtained by importing the reference's generator (tests/golden/static_twiddles.npz),
  * descriptor arithmetic known-answer test (test/unit_test/descriptor.cpp:76-109),
  * invalid-descriptor table (test/unit_test/instantiate_fft_tests.hpp:322-373),
  * planner level table (computed from prepare_implementation with the reference defaults).
"""
import numpy as np
import pytest

import helpers as H


def _cases(golden):
    out = []
    for k in golden["fft"].files:
        if k.endswith("_in"):
            prec, b, dims = k[:-3].split("_")
            out.append((k[:-3], prec, int(b[1:]), [int(x) for x in dims.split("x")]))
    return out


def test_static_twiddle_table_matches_reference_generator(oracle, golden):
    re, im = golden["tw"]["re"], golden["tw"]["im"]
    assert re.shape == (65, 65)
    L = oracle.lib()
    for n in range(65):
        for k in range(65):
            assert abs(L.pfo_static_twiddle(n, k, 0) - re[n, k]) <= 2.3e-16, (n, k)
            assert abs(L.pfo_static_twiddle(n, k, 1) - im[n, k]) <= 2.3e-16, (n, k)


@pytest.mark.parametrize("sg", [32, 64])
def test_oracle_matches_golden_vectors(oracle, golden, sg):
    """forward: out == golden out; backward: feeding golden out returns N * golden in (unnormalised backward,
    reference_data_wrangler.hpp:202-210)."""
    worst = {np.dtype(np.complex64): 0.0, np.dtype(np.complex128): 0.0}
    for key, prec, batch, dims in _cases(golden):
        x = golden["fft"][key + "_in"]
        y = golden["fft"][key + "_out"]
        n = int(np.prod(dims))
        desc = oracle.make_desc(dims, prec, batch=batch)
        out = oracle.compute(desc, oracle.FORWARD, x.ravel(), sg=sg, threads=4).reshape(y.shape)
        back = oracle.compute(desc, oracle.BACKWARD, y.ravel(), sg=sg, threads=4).reshape(x.shape)
        for b in range(batch):
            e1 = H.rel_l2(out[b], y[b])
            e2 = H.rel_l2(back[b], x[b].astype(np.complex128) * n)
            worst[x.dtype] = max(worst[x.dtype], e1, e2)
            assert e1 <= H.REL_L2_TOL[x.dtype] and e2 <= H.REL_L2_TOL[x.dtype], (key, b, e1, e2)
        assert H.check_reference_rule(out, y, n), key
    print("worst rel-L2", worst)


def test_oracle_global_level_matches_golden_vectors(oracle, golden):
    """the oracle's GLOBAL level (multi-factor algorithm, dispatcher/global_dispatcher.hpp:343-408) pinned to committed
    vectors at the reference's own GLOBAL sizes: GlobalTest 32768 / 65536 / 131072 in float, WorkgroupOrGlobal 8192 /
    16384 in double (instantiate_fft_tests.hpp:140-151) -- forward and unnormalised backward"""
    seen = []
    for key, prec, n, x, y in H.golden_global_cases(golden):
        desc = oracle.make_desc([n], prec)
        out = oracle.compute(desc, oracle.FORWARD, x.ravel(), threads=4).reshape(y.shape)
        back = oracle.compute(desc, oracle.BACKWARD, y.ravel(), threads=4).reshape(x.shape)
        e1, e2 = H.rel_l2(out[0], y[0]), H.rel_l2(back[0], x[0].astype(np.complex128) * n)
        assert e1 <= H.REL_L2_TOL[x.dtype] and e2 <= H.REL_L2_TOL[x.dtype], (key, e1, e2)
        assert H.check_reference_rule(out, y, n), key
        seen.append(key)
    assert seen == ["f32_b1_131072", "f32_b1_32768", "f32_b1_65536", "f64_b1_16384", "f64_b1_8192"], seen


def test_config1_values(oracle, golden):
    """BASELINE config 1 / SURVEY 8(c): fp32 N=64 batch=1, first input and output values"""
    x = golden["fft"]["f32_b1_64_in"][0]
    y = golden["fft"]["f32_b1_64_out"][0]
    np.testing.assert_allclose(x[:3], [0.1373785 - 0.19363792j, -0.5303392 - 0.7960795j, -0.20715714 + 0.8369587j],
                               rtol=1e-6)
    np.testing.assert_allclose(y[:3], [-3.6417406 + 3.9462976j, -3.398643 + 1.3439018j, -5.2978406 + 2.0263965j],
                               rtol=1e-6)
    out = oracle.dft_1d(x)
    assert H.rel_l2(out, y) < 1e-6


@pytest.mark.parametrize("level,sizes", [(0, [2, 3, 4, 5, 8, 9, 16]), (1, [32, 64, 96, 100, 128, 512]),
                                         (2, [256, 1024, 2048, 3072, 4096]), (3, [4096, 9800, 32768])])
def test_every_tier_against_numpy(oracle, level, sizes):
    rng = np.random.default_rng(level)
    for n in sizes:
        x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
        y = oracle.dft_1d(x, level=level)
        assert H.rel_l2(y, np.fft.fft(x.astype(np.complex128))) < 2e-6, (level, n)
        yb = oracle.dft_1d(x, direction=oracle.BACKWARD, level=level)
        assert H.rel_l2(yb, np.fft.ifft(x.astype(np.complex128)) * n) < 2e-6, (level, n)


def test_descriptor_buffer_count_kat(oracle):
    d = oracle.make_desc([2, 3], batch=2, fwd_strides=[8, 3], bwd_strides=[2, 4], fwd_distance=15, bwd_distance=1,
                         fwd_offset=3, bwd_offset=5)
    L = oracle.lib()
    assert L.pfo_input_count(d, 0) == 33 and L.pfo_output_count(d, 0) == 17
    assert L.pfo_input_count(d, 1) == 17 and L.pfo_output_count(d, 1) == 33


def _desc_from_case(oracle, case):
    name, lengths, fs, bs, fd, bd, batch, place = case

    def dist(strides, d):
        if d is not None:
            return d
        if strides is None:
            return None
        return int(np.prod([l * s for l, s in zip(lengths, strides)]))

    return oracle.make_desc(lengths, batch=batch, placement=place, fwd_strides=fs, bwd_strides=bs,
                            fwd_distance=dist(fs, fd), bwd_distance=dist(bs, bd))


@pytest.mark.parametrize("case", H.INVALID_CASES, ids=[c[0] for c in H.INVALID_CASES])
def test_invalid_descriptors_are_rejected(oracle, case):
    st, msg = oracle.validate(_desc_from_case(oracle, case))
    assert st == oracle.INVALID, (case[0], st, msg)


def test_valid_strided_layouts_are_accepted(oracle):
    for lengths, fs, bs, fd, bd in H.STRIDED_OOP_CASES + H.STRIDED_OOP_BATCH_INTERLEAVED_LIKE + H.STRIDED_IP_CASES:
        for batch in (1, 3, 33):
            if (fd == 1 or bd == 1) and batch > min(fs[0], bs[0]):
                continue
            case = ("ok", lengths, fs, bs, fd, bd, batch, 1 if fs != bs or fd != bd else 0)
            st, msg = oracle.validate(_desc_from_case(oracle, case))
            assert st == oracle.OK, (lengths, fs, bs, fd, bd, batch, msg)


def test_unsupported(oracle):
    st, _ = oracle.validate(oracle.make_desc([64], domain=0))
    assert st == oracle.UNSUPPORTED
    st, _ = oracle.validate(oracle.make_desc([4, 4], batch=2, fwd_strides=[8, 2], fwd_distance=64))
    assert st == oracle.UNSUPPORTED  # N-D with non-default layout
    st, _ = oracle.validate(oracle.make_desc([4096], fwd_strides=[2], fwd_distance=8192))
    assert st == oracle.UNSUPPORTED  # UNPACKED that does not fit a sub-group


def test_planner_level_table(oracle):
    """levels chosen by prepare_implementation with the reference defaults"""
    t = [(64, 4, 32, oracle.SUBGROUP, [2, 32]), (64, 4, 64, oracle.SUBGROUP, [1, 64]),
         (1024, 4, 32, oracle.WORKGROUP, [1, 32, 1, 32]), (1024, 4, 64, oracle.SUBGROUP, [16, 64]),
         (4096, 4, 32, oracle.WORKGROUP, [2, 32, 2, 32]), (4096, 4, 64, oracle.WORKGROUP, [1, 64, 1, 64]),
         (16, 4, 32, oracle.WORKITEM, []), (16, 8, 32, oracle.SUBGROUP, [1, 16]), (8, 8, 32, oracle.WORKITEM, [])]
    for n, sb, sg, level, factors in t:
        impl = oracle.prepare_implementation(n, sb, sg)
        assert impl.level == level, (n, sb, sg, impl.level)
        assert [impl.factors[0][i] for i in range(impl.n_factors[0])] == factors or level == oracle.WORKITEM
    assert oracle.prepare_implementation(1 << 20, 8, 32).level == oracle.GLOBAL
    with pytest.raises(oracle.OracleError):
        oracle.prepare_implementation(4099, 4, 32)  # large prime


def test_layouts_scales_offsets_nd(oracle):
    """the oracle honours strides / distances / offsets / scales / split storage / N-D like the reference"""
    rng = np.random.default_rng(5)
    # batch interleaved in, packed out, forward scale 2, offsets
    n, b = 64, 5
    x = (rng.uniform(-1, 1, (b, n)) + 1j * rng.uniform(-1, 1, (b, n))).astype(np.complex64)
    d = oracle.make_desc([n], batch=b, fwd_strides=[b], fwd_distance=1, fwd_offset=3, bwd_offset=7, fwd_scale=2.0)
    buf = H.scatter(x, [b], 1, 3, oracle.lib().pfo_input_count(d, 0))
    out = oracle.compute(d, 0, buf)
    got = H.gather(out, b, [n], [1], n, 7)
    assert H.rel_l2(got, 2.0 * np.fft.fft(x.astype(np.complex128), axis=1)) < 2e-6
    assert np.all(out[:7] == 0)
    # N-D, backward with scale, split storage
    dims = [4, 6, 8]
    x = (rng.uniform(-1, 1, [2] + dims) + 1j * rng.uniform(-1, 1, [2] + dims))
    d = oracle.make_desc(dims, "f64", batch=2, storage=1, bwd_scale=0.5)
    re, im = np.ascontiguousarray(x.real.ravel()), np.ascontiguousarray(x.imag.ravel())
    ore, oim = np.zeros_like(re), np.zeros_like(im)
    oracle.compute(d, 1, re, ore, im, oim)
    ref = 0.5 * np.fft.ifftn(x, axes=(1, 2, 3)) * np.prod(dims)
    assert H.rel_l2(ore + 1j * oim, ref.ravel()) < 1e-14
