"""Full-size BASELINE configs 3 and 5, the two-pass 2-D plan, and buffers beyond 4 GiB -- through the C ABI on an MI355X.

Full sizes cannot be compared element by element on the host in reasonable time, so they are checked through
size-independent properties (Parseval on every batch, forward->backward round trip) plus sampled transforms against
NumPy in double precision.  Tolerances: rel-L2 per transform <= helpers.REL_L2_TOL (2e-6 fp32 / 5e-15 fp64), the same
bar as tests/test_gpu_parity.py, far inside BASELINE.json's 1e-4.
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _mods():
    import gpu_utils as G
    import portfft_amd as pf
    return G, pf, G.torch_mod()


def _full_size_properties(lengths, batch, prec, samples):
    G, pf, torch = _mods()
    cdt = torch.complex64 if prec == "f32" else torch.complex128
    tol = H.REL_L2_TOL[np.dtype(np.complex64 if prec == "f32" else np.complex128)]
    n = int(np.prod(lengths))
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.empty(batch * n, dtype=cdt, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1, generator=g)
    y = torch.empty_like(x)
    plan = G.make_descriptor(lengths, prec, batch=batch).commit()
    plan.compute_forward(x, y).wait()
    xs, ys = x.view(batch, n), y.view(batch, n)
    for b in samples:
        ref = np.fft.fftn(xs[b].cpu().numpy().astype(np.complex128).reshape(lengths)).ravel()
        assert H.rel_l2(ys[b].cpu().numpy(), ref) <= tol, (lengths, prec, "batch", b)
    # Parseval on every batch: sum |X|^2 == n * sum |x|^2
    ex = (xs.abs().double() ** 2).sum(dim=1)
    ey = (ys.abs().double() ** 2).sum(dim=1)
    assert float(((ey / (n * ex)) - 1).abs().max()) < (1e-5 if prec == "f32" else 1e-12)
    # round trip on every batch: backward(forward(x)) == n * x  (x is overwritten by its reconstruction)
    z = torch.empty_like(x)
    plan.compute_backward(y, z).wait()
    err = (z.view(batch, n) / n - xs).abs().double().pow(2).sum(dim=1).sqrt() / ex.sqrt()
    assert float(err.max()) <= tol, float(err.max())
    return plan


def test_full_size_config3_properties():
    """BASELINE configs[2]: fp64 C2C 1-D N=2^20 batch=128 (GLOBAL tier, two launches through scratch)."""
    plan = _full_size_properties([1 << 20], 128, "f64", [0, 1, 63, 127])
    assert plan.info().dims[0].tier == 3


def test_full_size_config5_properties():
    """BASELINE configs[4]: fp32 C2C 2-D 1024 x 1024 batch=256 (two-pass 2-D plan)."""
    _full_size_properties([1024, 1024], 256, "f32", [0, 1, 128, 255])


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_two_pass_2d_plan(prec):
    """The two-pass 2-D plan (rows + first column radix, then short batch-interleaved columns) against NumPy and
    against the per-dimension plan it replaces (PFFT_2D_TWO_PASS=0): both placements, offsets, scales, both
    directions, leading dimensions, an OUT_OF_PLACE plan executed with aliasing buffers."""
    G, pf, torch = _mods()
    dtype = np.complex64 if prec == "f32" else np.complex128
    # ... plus shapes whose column dimension is too long for one wide column pass (two column-shaped stages through
    # scratch, plan_batch_interleaved_two_stage with an outer index) and a runtime-planned row length
    shapes = ([256, 256], [64, 1024], [1024, 1024], [512, 512], [16, 2048], [1000, 1024], [3, 128, 512], [2, 2, 64, 256],
              [4096, 64], [2048, 1536], [2, 4096, 48], [1080, 1920])
    for dims in shapes:
        n = int(np.prod(dims))
        batch = 3 if n <= (1 << 18) else 2
        x, y = H.gen_fourier_data(batch, dims, dtype, seed=n % 1000)
        for place, storage in ((0, 0), (1, 0), (0, 1), (1, 1)):  # storage 1: SPLIT_COMPLEX planes
            d = G.make_descriptor(dims, prec, batch=batch, placement=place, storage=storage, fwd_scale=0.25,
                                  bwd_scale=2.0, fwd_offset=7, bwd_offset=7 if place == 0 else 13)
            got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
            for b in range(batch):
                assert H.rel_l2(got[b], 0.25 * y[b]) <= H.REL_L2_TOL[np.dtype(dtype)], ("fwd", dims, place, storage, b)
            back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
            for b in range(batch):
                assert H.rel_l2(back[b], 2.0 * n * x[b].astype(np.complex128)) <= H.REL_L2_TOL[np.dtype(dtype)], \
                    ("bwd", dims, place, storage, b)
        # the plan it replaces gives the same answer up to rounding
        d = G.make_descriptor(dims, prec, batch=batch)
        got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
        os.environ["PFFT_2D_TWO_PASS"] = "0"
        try:
            old, _ = G.transform_packed(G.make_descriptor(dims, prec, batch=batch), pf.direction.FORWARD, x)
        finally:
            del os.environ["PFFT_2D_TWO_PASS"]
        assert H.rel_l2(got, old) <= 2 * H.REL_L2_TOL[np.dtype(dtype)], ("vs per-dimension plan", dims)
        # OUT_OF_PLACE descriptor, aliasing buffers (the reference's in-place overload forwards to the out-of-place
        # one: committed_descriptor.hpp:171-176)
        plan = G.make_descriptor(dims, prec, batch=batch).commit()
        buf = torch.from_numpy(x.ravel().copy()).cuda()
        plan.compute_forward(buf).wait()
        assert H.rel_l2(buf.cpu().numpy().reshape(batch, -1), y.reshape(batch, -1)) <= H.REL_L2_TOL[np.dtype(dtype)], \
            ("aliased", dims)
    for storage in (0, 1):
        plan = G.make_descriptor([1024, 1024], prec, batch=2, storage=storage).commit()
        info = plan.info()
        assert info.dims[1].tier == 1 and info.dims[0].tier == 1
        assert int(np.prod(list(info.dims[0].factors)[:info.dims[0].n_factors])) == 1024
        assert info.dims[0].factors[0] in (2, 4, 8), storage  # the column radix fused into pass 1


def _big_case(n, batch, prec="f32", layout_in="P", layout_out="P", split=False):
    """device-generated data, transforms around the 2^31-element and 2^32-byte marks checked against NumPy"""
    G, pf, torch = _mods()
    cdt = torch.complex64 if prec == "f32" else torch.complex128
    rdt = torch.float32 if prec == "f32" else torch.float64
    total = n * batch
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.view_as_complex(torch.rand(total, 2, dtype=rdt, device="cuda", generator=g) * 2 - 1)
    d = pf.descriptor([n], prec)
    d.number_of_transforms = batch
    if layout_in == "BI":
        d.forward_strides, d.forward_distance = [batch], 1
    if layout_out == "BI":
        d.backward_strides, d.backward_distance = [batch], 1
    if split:
        d.complex_storage = pf.complex_storage.SPLIT_COMPLEX
    plan = d.commit()
    if split:
        a, b = x.real.contiguous(), x.imag.contiguous()
        yr = torch.empty(total, dtype=rdt, device="cuda")
        yi = torch.empty(total, dtype=rdt, device="cuda")
        plan.compute_forward(a, b, yr, yi).wait()
    else:
        y = torch.empty(total, dtype=cdt, device="cuda")
        plan.compute_forward(x, y).wait()
    esz = 8 if prec == "f32" else 16
    marks = {0, 1, batch // 2, batch - 2, batch - 1}
    for m in ((1 << 32) // (esz * n) + 3, (1 << 31) // n + 1):
        if m < batch:
            marks.add(m)
    tol = 2e-6 if prec == "f32" else 1e-14
    for b_ in sorted(marks):
        xi_ = x[b_::batch][:n] if layout_in == "BI" else x[b_ * n:(b_ + 1) * n]
        if split:
            yo = (torch.complex(yr[b_::batch][:n], yi[b_::batch][:n]) if layout_out == "BI"
                  else torch.complex(yr[b_ * n:(b_ + 1) * n], yi[b_ * n:(b_ + 1) * n]))
        else:
            yo = y[b_::batch][:n] if layout_out == "BI" else y[b_ * n:(b_ + 1) * n]
        ref = np.fft.fft(xi_.cpu().numpy().astype(np.complex128))
        assert H.rel_l2(yo.cpu().numpy(), ref) <= tol, (n, batch, prec, layout_in, layout_out, split, b_)
    del x
    torch.cuda.empty_cache()


def test_buffers_beyond_4gib():
    """Every tier addresses buffers beyond 4 GiB / 2^31 elements with 64-bit offsets (2.2-9.6 GiB per buffer)."""
    _big_case(4096, 300000)                                       # work-group tier, 9.2 GiB
    _big_case(16, 80000000)                                       # register tier, 9.5 GiB
    _big_case(4096, 300000, split=True)                           # split storage
    _big_case(4096, 300000, layout_in="BI", layout_out="BI")      # two column-shaped stages through scratch, BIG forms (9.2 GiB)
    # (round 6: a batch-interleaved array of 4 GiB and more keeps the two-stage plan -- 64-bit butterfly-leg offsets,
    #  stockham_strided.hpp strided_io_big -- instead of falling to narrow groups / the generic tier: 0.08 -> 0.34 of the HBM peak)
    G, pf, _ = _mods()
    d = pf.descriptor([4096], "f32")
    d.number_of_transforms = 140000
    d.forward_strides, d.forward_distance, d.backward_strides, d.backward_distance = [140000], 1, [140000], 1
    dim = d.commit().info().dims[0]
    assert dim.tier == 3 and list(dim.factors[:2]) == [128, 32], (dim.tier, list(dim.factors[:2]))
    os.environ["PFFT_NO_BIG_BI"] = "1"
    try:
        assert d.commit().info().dims[0].tier != 3, "the round-5 twin: no two-stage plan at 4 GiB and beyond"
    finally:
        del os.environ["PFFT_NO_BIG_BI"]
    # (the round's last session: lengths of 1025 ... 2048 points keep their ONE-pass plan on the wide register-resident group --
    #  the BIG form of stockham_strided_hx_kernel -- at 4 GiB and beyond too: 0.33 -> 0.47; an unaligned batch count on default policies)
    _big_case(2048, 300000, layout_in="BI", layout_out="BI")      # 4.9 GiB per buffer
    d = pf.descriptor([2048], "f32")
    d.number_of_transforms = 300000
    d.forward_strides, d.forward_distance, d.backward_strides, d.backward_distance = [300000], 1, [300000], 1
    dim = d.commit().info().dims[0]
    assert dim.tier == 1 and dim.ffts_per_workgroup == 16, (dim.tier, dim.ffts_per_workgroup)
    _big_case(2048, 270003, layout_in="BI", layout_out="BI")
    _big_case(512, 2400000, layout_in="BI")                       # strided tier, row-shaped output
    _big_case(1200, 1000000)                                      # runtime-specialised length
    _big_case(1 << 20, 1200)                                      # GLOBAL tier fp32, chunked scratch
    _big_case(65536, 10000, prec="f64")                           # GLOBAL tier fp64
    # intermediates of 128-256 MiB: the cache-policy twins (writer / reader) without chunking
    _big_case(4096, 6000, layout_in="BI", layout_out="BI")        # two column-shaped stages, 188 MiB of scratch
    _big_case(65536, 400)                                         # four-step fp32, 200 MiB
    _big_case(1 << 20, 12, prec="f64")                            # four-step fp64, 192 MiB


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_cross_lane_variants(prec):
    """The cross-lane sub-group kernels (stockham_xlane.hpp: in-wave DPP / ds_swizzle transpose instead of the LDS
    exchange; reference role: common/subgroup.hpp:141-216) against NumPy and against the LDS-staged kernels they
    stand next to, both directions, ragged batch counts, scale."""
    G, pf, torch = _mods()
    dtype = np.complex64 if prec == "f32" else np.complex128
    tol = H.REL_L2_TOL[np.dtype(dtype)]
    for n in ((16, 64, 256) if prec == "f32" else (64, 256)):
        for batch in (1, 33, 1000):
            x, y = H.gen_fourier_data(batch, [n], dtype, seed=n + batch)
            base, _ = G.transform_packed(G.make_descriptor([n], prec, batch=batch, fwd_scale=0.5), pf.direction.FORWARD, x)
            os.environ["PFFT_XLANE"] = "1"
            try:
                d = G.make_descriptor([n], prec, batch=batch, fwd_scale=0.5)
                got, _ = G.transform_packed(d, pf.direction.FORWARD, x)
                back, _ = G.transform_packed(d, pf.direction.BACKWARD, y)
            finally:
                del os.environ["PFFT_XLANE"]
            for b in range(batch):
                assert H.rel_l2(got[b], 0.5 * y[b]) <= tol, ("xlane fwd", prec, n, batch, b)
                assert H.rel_l2(back[b], n * x[b].astype(np.complex128)) <= tol, ("xlane bwd", prec, n, batch, b)
            assert H.rel_l2(got, base) <= 2 * tol, ("xlane vs staged", prec, n, batch)


def test_python_mirror_events_dependencies_and_argument_checks():
    """portfft_amd/api.py: per-submission events, `dependencies` (this module's events and torch.cuda.Event across
    streams), copies of a committed descriptor, and the argument checks in front of the C ABI (a tensor on the wrong
    device / of the wrong dtype / too short / not contiguous raises invalid_configuration instead of faulting)."""
    G, pf, torch = _mods()
    n, batch = 4096, 64
    x, y = H.gen_fourier_data(batch, [n], np.complex64, seed=5)
    d = G.make_descriptor([n], "f32", batch=batch)
    side = torch.cuda.Stream()
    plan = d.commit()
    xin = torch.empty(batch * n, dtype=torch.complex64, device="cuda")
    out = torch.empty_like(xin)
    with torch.cuda.stream(side):  # the input is produced on another stream; only the dependency orders the FFT
        xin.copy_(torch.from_numpy(x.ravel()), non_blocking=False)
        produced = torch.cuda.Event()
        produced.record(side)
    ev = plan.compute_forward(xin, out, dependencies=[produced])
    assert isinstance(ev, pf.event)
    ev.wait()
    assert ev.is_complete()
    assert H.rel_l2(out.cpu().numpy().reshape(batch, n), y) <= 2e-6
    # chained through this module's events, then a copy of the plan (own scratch, shared twiddles)
    back = torch.empty_like(xin)
    e2 = plan.compute_backward(out, back, dependencies=[ev])
    clone = plan.copy()
    del plan
    out2 = torch.empty_like(xin)
    e3 = clone.compute_forward(xin, out2, dependencies=[e2])
    e3.wait()
    assert torch.equal(out, out2)
    assert H.rel_l2(back.cpu().numpy().reshape(batch, n), n * x.astype(np.complex128)) <= 2e-6
    # argument checks
    for bad, what in ((torch.empty(batch * n, dtype=torch.complex64), "device memory"),
                      (torch.empty(batch * n, dtype=torch.complex128, device="cuda"), "dtype"),
                      (torch.empty(batch * n - 1, dtype=torch.complex64, device="cuda"), "elements"),
                      (torch.empty(2 * batch * n, dtype=torch.complex64, device="cuda")[::2], "contiguous")):
        with pytest.raises(pf.invalid_configuration, match=what):
            clone.compute_forward(bad, out2)
    with pytest.raises(pf.invalid_configuration):
        clone.compute_forward(xin, out2, out2)  # not one of the overloads
    # a float32 view of interleaved data is accepted (two scalars per element)
    clone.compute_forward(torch.view_as_real(xin).reshape(-1), torch.view_as_real(out2).reshape(-1)).wait()
    assert torch.equal(out, out2)


def test_two_pass_2d_plan_of_split_storage_in_cache_sized_chunks():
    """BASELINE config 5's shape in SPLIT_COMPLEX storage (round 6): from 128 MiB of data the two-pass 2-D plan of split planes
    runs chunk by chunk on the writer / reader twins of its two registered kernels like the interleaved plan does
    (plan_nd.cpp; the reference has one code path for both storages, committed_descriptor_impl.hpp:899-950).  A batch that leaves a
    ragged last chunk, out of place and in place (a chunk then goes through the scratch's two halves), both directions,
    against NumPy on sampled matrices and bit for bit against the streamed one-launch-per-pass plan
    (PFFT_NO_SPLIT_2D_CACHED=1)."""
    G, pf, torch = _mods()

    def run_case(lengths, batch, prec, placement, env=None):
        rt = torch.float32 if prec == "f32" else torch.float64
        n = int(np.prod(lengths))
        g = torch.Generator(device="cuda").manual_seed(23)
        xr = torch.empty(batch * n, dtype=rt, device="cuda").uniform_(-1, 1, generator=g)
        xi = torch.empty(batch * n, dtype=rt, device="cuda").uniform_(-1, 1, generator=g)
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            plan = G.make_descriptor(lengths, prec, batch=batch, placement=placement, storage=1).commit()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if placement == 0:
            yr, yi = xr.clone(), xi.clone()
            plan.compute_forward(yr, yi).wait()
        else:
            yr, yi = torch.empty_like(xr), torch.empty_like(xi)
            plan.compute_forward(xr, xi, yr, yi).wait()
        return xr, xi, yr, yi, plan

    tol = {"f32": 2e-6, "f64": 5e-15}
    for lengths, batch, prec in (([1024, 1024], 37, "f32"), ([1024, 1024], 21, "f64")):
        n = int(np.prod(lengths))
        for placement in (1, 0):
            xr, xi, yr, yi, plan = run_case(lengths, batch, prec, placement)
            assert list(plan.info().launches)[0] > 2, "several chunks of two launches"
            for b in (0, batch // 2, batch - 1):
                x = (xr.view(batch, n)[b].cpu().numpy().astype(np.float64) + 1j * xi.view(batch, n)[b].cpu().numpy()).reshape(lengths)
                ref = np.fft.fftn(x).ravel()
                got = yr.view(batch, n)[b].cpu().numpy().astype(np.float64) + 1j * yi.view(batch, n)[b].cpu().numpy()
                assert H.rel_l2(got, ref) <= tol[prec], (lengths, batch, prec, placement, b)
            _, _, y0r, y0i, plan0 = run_case(lengths, batch, prec, placement, {"PFFT_NO_SPLIT_2D_CACHED": "1"})
            assert list(plan0.info().launches)[0] == 2, "the streamed twin: one launch per pass"
            assert torch.equal(yr, y0r) and torch.equal(yi, y0i), (lengths, batch, prec, placement)
            zr, zi = torch.empty_like(xr), torch.empty_like(xi)
            plan.compute_backward(yr, yi, zr, zi).wait()
            err = float((((zr / n - xr).double().pow(2).sum() + (zi / n - xi).double().pow(2).sum()) /
                         (xr.double().pow(2).sum() + xi.double().pow(2).sum())).sqrt())
            assert err <= tol[prec], (lengths, batch, prec, placement, err)
            del yr, yi, y0r, y0i, zr, zi
        torch.cuda.empty_cache()


def test_batch_interleaved_at_batch_counts_that_are_no_multiple_of_a_line():
    """BATCH_INTERLEAVED with a batch count that is no multiple of 16 fp32 / 8 fp64 transforms (the reference's own test batch,
    33000, is one: instantiate_fft_tests.hpp) -- every row pitch is unaligned, every 128-byte segment of a group shares its first
    and last line with the neighbouring group.  From 64 MiB of data such stages run kernels compiled at commit on default
    cache policies with the XCD-contiguous walk (kernels.hpp aux_of_policy, policy 3; round 6: N = 1024 at a batch of 131 077
    0.20 -> 0.50 of the HBM peak), pre-compiled lengths included, the two-stage plan of long transforms too.  Against NumPy
    on sampled transforms, both directions, and against the streamed twin (PFFT_NO_UNALIGNED_POLICY=1)."""
    G, pf, torch = _mods()

    def run_case(n, batch, prec, env=None):
        cdt = torch.complex64 if prec == "f32" else torch.complex128
        g = torch.Generator(device="cuda").manual_seed(n + batch)
        x = torch.empty(batch * n, dtype=cdt, device="cuda")
        torch.view_as_real(x).uniform_(-1, 1, generator=g)
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            d = pf.descriptor([n], prec)
            d.number_of_transforms = batch
            d.forward_strides, d.forward_distance, d.backward_strides, d.backward_distance = [batch], 1, [batch], 1
            plan = d.commit()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        y = torch.empty_like(x)
        plan.compute_forward(x, y).wait()
        return x, y, plan

    tol = {"f32": 2e-6, "f64": 5e-15}
    # ((2048, 8195): the wide register-resident group of the round's last session on policy 3)
    for n, batch, prec in ((1024, 16391, "f32"), (768, 21851, "f32"), (256, 32771, "f64"), (4096, 4099, "f32"), (1000, 16387, "f64"),
                           (2048, 8195, "f32")):
        x, y, plan = run_case(n, batch, prec)
        for b in (0, 1, batch // 2, batch - 1):
            ref = np.fft.fft(x[b::batch].cpu().numpy().astype(np.complex128))
            assert H.rel_l2(y[b::batch].cpu().numpy(), ref) <= tol[prec], (n, batch, prec, b)
        _, y0, plan0 = run_case(n, batch, prec, {"PFFT_NO_UNALIGNED_POLICY": "1"})
        assert plan0.info().knob_mask != plan.info().knob_mask
        diff = float(((y - y0).abs().double().pow(2).sum() / y0.abs().double().pow(2).sum()).sqrt())
        assert diff <= tol[prec], (n, batch, prec, "against the streamed twin", diff)
        z = torch.empty_like(x)
        plan.compute_backward(y, z).wait()
        err = float(((z / n - x).abs().double().pow(2).sum() / x.abs().double().pow(2).sum()).sqrt())
        assert err <= tol[prec], (n, batch, prec, "round trip", err)
        del x, y, y0, z
    # SPLIT_COMPLEX planes at such a batch count (a plane's pitch: batch * sizeof(scalar)): the wide groups of N = 513 ... 2048 take policy 3 too
    # (and, N <= 512, the LDS-resident kernels compiled at commit: (256, 32771))
    for n, batch, prec in ((768, 21851, "f32"), (2048, 4099, "f64"), (256, 32771, "f32")):
        rdt = torch.float32 if prec == "f32" else torch.float64
        g = torch.Generator(device="cuda").manual_seed(n + batch)
        xr = torch.empty(batch * n, dtype=rdt, device="cuda").uniform_(-1, 1, generator=g)
        xi = torch.empty(batch * n, dtype=rdt, device="cuda").uniform_(-1, 1, generator=g)
        outs = []
        for env in ({}, {"PFFT_NO_UNALIGNED_POLICY": "1"}):
            os.environ.update(env)
            try:
                d = pf.descriptor([n], prec)
                d.number_of_transforms = batch
                d.complex_storage = pf.complex_storage.SPLIT_COMPLEX
                d.forward_strides, d.forward_distance, d.backward_strides, d.backward_distance = [batch], 1, [batch], 1
                plan = d.commit()
            finally:
                for k in env:
                    del os.environ[k]
            yr, yi = torch.empty_like(xr), torch.empty_like(xi)
            plan.compute_forward(xr, xi, yr, yi).wait()
            outs.append((torch.complex(yr, yi), plan.info().knob_mask))
        assert outs[0][1] != outs[1][1]
        y = outs[0][0]
        for b in (0, 1, batch // 2, batch - 1):
            ref = np.fft.fft(torch.complex(xr[b::batch], xi[b::batch]).cpu().numpy().astype(np.complex128))
            assert H.rel_l2(y[b::batch].cpu().numpy(), ref) <= tol[prec], ("split", n, batch, prec, b)
        diff = float(((y - outs[1][0]).abs().double().pow(2).sum() / y.abs().double().pow(2).sum()).sqrt())
        assert diff <= tol[prec], ("split", n, batch, prec, "against the streamed twin", diff)
        del xr, xi, y, outs
    torch.cuda.empty_cache()


def test_cache_sized_chunks_and_policy_twins():
    """The two-launch plans run chunk by chunk (256 MiB of intermediate per chunk) on the writer / reader cache-policy
    twins once the intermediate reaches 128 MiB: batch counts that leave a ragged last chunk, both directions, in-place
    (the 2-D plan then routes a chunk through its scratch), against NumPy on sampled transforms and against the same
    plan with PFFT_CACHE_CHUNK_MIB=0 (everything streamed, one launch per pass) bit for bit."""
    G, pf, torch = _mods()

    def run_case(lengths, batch, prec, placement, env=None):
        cdt = torch.complex64 if prec == "f32" else torch.complex128
        n = int(np.prod(lengths))
        g = torch.Generator(device="cuda").manual_seed(11)
        x = torch.empty(batch * n, dtype=cdt, device="cuda")
        torch.view_as_real(x).uniform_(-1, 1, generator=g)
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            plan = G.make_descriptor(lengths, prec, batch=batch, placement=placement).commit()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if placement == 0:
            y = x.clone()
            plan.compute_forward(y).wait()
        else:
            y = torch.empty_like(x)
            plan.compute_forward(x, y).wait()
        return x, y, plan

    tol = {"f32": 2e-6, "f64": 5e-15}
    for lengths, batch, prec in (([1024, 1024], 37, "f32"), ([512, 2048], 70, "f32"), ([65536], 700, "f32"),
                                 ([1 << 18], 150, "f64"), ([1 << 20], 40, "f32"), ([1 << 20], 37, "f64")):
        n = int(np.prod(lengths))
        for placement in (1, 0):
            x, y, plan = run_case(lengths, batch, prec, placement)
            for b in (0, batch // 2, batch - 1):
                ref = np.fft.fftn(x.view(batch, n)[b].cpu().numpy().astype(np.complex128).reshape(lengths)).ravel()
                assert H.rel_l2(y.view(batch, n)[b].cpu().numpy(), ref) <= tol[prec], (lengths, batch, prec, placement, b)
            # same kernels, same arithmetic, other cache policy and launch structure: bit-identical
            _, y0, _ = run_case(lengths, batch, prec, placement, {"PFFT_CACHE_CHUNK_MIB": "0"})
            assert torch.equal(y, y0), (lengths, batch, prec, placement)
            # consecutive chunks overlap by default (the first launch of chunk c + 1 carries no in-order barrier and
            # the scratch of the four-step plans is double-buffered): strictly in order and with a second stream the
            # results are the same bits
            for mode in ("0", "1"):
                _, y1, _ = run_case(lengths, batch, prec, placement, {"PFFT_CHUNK_OVERLAP": mode})
                assert torch.equal(y, y1), (lengths, batch, prec, placement, "overlap mode", mode)
                del y1
            # backward over the chunked plan restores the input
            z = torch.empty_like(x)
            plan.compute_backward(y, z).wait()
            err = float(((z / n - x).abs().double().pow(2).sum() / x.abs().double().pow(2).sum()).sqrt())
            assert err <= tol[prec], (lengths, batch, prec, placement, err)
            del y, y0, z
        torch.cuda.empty_cache()
    # a chunked execute captured into a HIP graph (in-order chain of kernel nodes) and replayed on new data
    lengths, batch, prec = [1024, 1024], 37, "f32"
    n = int(np.prod(lengths))
    x, y, plan = run_case(lengths, batch, prec, 1)
    assert list(plan.info().launches) == [4, 4]  # 37 matrices of 8 MiB: chunks of 32 + 5, two launches each
    _, _, plan0 = run_case(lengths, batch, prec, 1, {"PFFT_CACHE_CHUNK_MIB": "0"})
    assert list(plan0.info().launches) == [2, 2]
    s1 = torch.cuda.Stream()
    plan_s = G.make_descriptor(lengths, prec, batch=batch).commit(s1)
    xin = torch.zeros_like(x)
    out = torch.empty_like(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):
        plan_s.compute_forward(xin, out, want_event=False)
    torch.cuda.synchronize()
    xin.copy_(x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, y), "graph replay of a chunked plan"


def test_four_step_stage_pairs_and_split_storage():
    """The four-step (GLOBAL) tier after round 3: registered stage pairs (group-major intermediate, tiled-input stage B,
    software-pipelined forms), lengths whose two factors have no pair (fall back to the default entries), cache-sized
    chunks with a ragged last chunk, SPLIT_COMPLEX data through the mixed-storage stages (stage B row-staged through LDS
    for power-of-two rows, writer / reader policies).  Against NumPy on sampled transforms, round trip through the
    backward plan, and against the same descriptor planned without the pairs (PFFT_NO_FS_PAIRS=1) / with the split plans
    streamed (PFFT_SPLIT_CACHED=0, PFFT_NO_MIXED_ROWS=1): same results within the tolerance."""
    G, pf, torch = _mods()

    def commit(lengths, prec, batch, storage, env=None):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            return G.make_descriptor(lengths, prec, batch=batch, storage=storage).commit()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v

    def forward(plan, x, split):
        if split:
            re, im = x.real.contiguous(), x.imag.contiguous()
            ore, oim = torch.empty_like(re), torch.empty_like(im)
            plan.compute_forward(re, im, ore, oim).wait()
            return torch.complex(ore, oim)
        y = torch.empty_like(x)
        plan.compute_forward(x, y).wait()
        return y

    def backward(plan, y, split):
        if split:
            re, im = y.real.contiguous(), y.imag.contiguous()
            ore, oim = torch.empty_like(re), torch.empty_like(im)
            plan.compute_backward(re, im, ore, oim).wait()
            return torch.complex(ore, oim)
        z = torch.empty_like(y)
        plan.compute_backward(y, z).wait()
        return z

    cases = [  # (n, prec, batch, storage)
        (1 << 15, "f32", 5, 0), (1 << 16, "f32", 5, 0), (1 << 17, "f32", 3, 0), (1 << 18, "f32", 3, 0),
        (1 << 19, "f32", 3, 0), (1 << 20, "f32", 2, 0), (1 << 21, "f32", 2, 0), (1 << 22, "f32", 2, 0),
        (1 << 16, "f64", 3, 0), (1 << 20, "f64", 2, 0),
        (1 << 16, "f32", 600, 0), (1 << 18, "f32", 150, 0), (1 << 20, "f64", 21, 0),   # chunked, ragged last chunk
        (1 << 16, "f32", 5, 1), (1 << 20, "f32", 2, 1), (1 << 18, "f64", 3, 1), (1000000, "f32", 2, 1),
        (1 << 16, "f32", 600, 1), (1 << 20, "f32", 40, 1),                              # split, chunked
    ]
    for n, prec, batch, storage in cases:
        cdt = torch.complex64 if prec == "f32" else torch.complex128
        tol = H.REL_L2_TOL[np.dtype(np.complex64 if prec == "f32" else np.complex128)]
        g = torch.Generator(device="cuda").manual_seed(n % 1000 + batch)
        x = torch.empty(batch * n, dtype=cdt, device="cuda")
        torch.view_as_real(x).uniform_(-1, 1, generator=g)
        split = storage == 1
        plan = commit([n], prec, batch, storage)
        y = forward(plan, x, split)
        for b in sorted({0, batch // 2, batch - 1}):
            ref = np.fft.fft(x.view(batch, n)[b].cpu().numpy().astype(np.complex128))
            assert H.rel_l2(y.view(batch, n)[b].cpu().numpy(), ref) <= tol, (n, prec, batch, storage, b)
        z = backward(plan, y, split)
        err = float(((z / n - x).abs().double().pow(2).sum() / x.abs().double().pow(2).sum()).sqrt())
        assert err <= tol, (n, prec, batch, storage, "round trip", err)
        env = {"PFFT_NO_FS_PAIRS": "1"} if not split else {"PFFT_SPLIT_CACHED": "0", "PFFT_NO_MIXED_ROWS": "1"}
        y0 = forward(commit([n], prec, batch, storage, env), x, split)
        diff = float(((y - y0).abs().double().pow(2).sum() / y0.abs().double().pow(2).sum()).sqrt())
        assert diff <= tol, (n, prec, batch, storage, "vs the round-2 plan", diff)
        if split:
            # group-major intermediate + tiled-input mixed stage B against the row-major / row-staged plan
            y1 = forward(commit([n], prec, batch, storage, {"PFFT_NO_SPLIT_TILED": "1"}), x, split)
            diff = float(((y - y1).abs().double().pow(2).sum() / y1.abs().double().pow(2).sum()).sqrt())
            assert diff <= tol, (n, prec, batch, "tiled against row-major split intermediate", diff)
            del y1
        if prec == "f64" and n == 1 << 20 and not split:
            # this pair's stage B carries the inter-stage twiddles on its loads: against the same pair with the
            # modifier on stage A's stores
            y1 = forward(commit([n], prec, batch, storage, {"PFFT_NO_LTW": "1"}), x, split)
            diff = float(((y - y1).abs().double().pow(2).sum() / y1.abs().double().pow(2).sum()).sqrt())
            assert diff <= tol, (n, prec, batch, "load-side against store-side modifier", diff)
            del y1
        del x, y, z, y0, plan


@pytest.mark.gpu
def test_three_stage_plan_for_very_long_transforms():
    """N > 2^22 (plan_global.cpp plan_three_stage): N = n1 * n2 * n3, the four-step applied twice -- S1 in place on the user's
    output buffer, S2 into tiles of the scratch, S3 (tiled-input stage B of n3 = 1024) to X[k1 + n1 k2 + n1 n2 k3];
    S2 / S3 chunk by chunk.  Factors through the plan info; against NumPy, round trip, in-place execution, a ragged last
    chunk, and the two-stage plan of the same descriptor (PFFT_NO_THREE_STAGE=1).  Also forced on a shorter length
    (PFFT_THREE_STAGE_MIN) so that a batch of several chunks stays small."""
    G, pf, torch = _mods()

    def commit(n, prec, batch, env=None, placement=None):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            d = G.make_descriptor([n], prec, batch=batch)
            if placement is not None:
                d.placement = placement
            return d.commit()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v

    cases = [  # (n, prec, batch, factors, env)
        (1 << 23, "f32", 2, [128, 256, 256], None), (3 << 22, "f32", 1, None, None), (5 << 20, "f32", 3, None, None), (1 << 24, "f32", 3, [256, 256, 256], None),
        (1 << 25, "f32", 1, [256, 512, 256], None), (1 << 23, "f64", 1, [64, 128, 1024], None),
        (1 << 27, "f32", 1, [256, 512, 1024], None),  # (no two-stage plan exists for this length)
        (1 << 22, "f32", 5, [128, 128, 256], {"PFFT_THREE_STAGE_MIN": "4194304", "PFFT_CACHE_CHUNK_MIB": "64"}),
        (5 << 20, "f64", 3, None, {"PFFT_THREE_STAGE_MIN": "4194304", "PFFT_CACHE_CHUNK_MIB": "96"}),
    ]
    for n, prec, batch, factors, env in cases:
        cdt = torch.complex64 if prec == "f32" else torch.complex128
        tol = H.REL_L2_TOL[np.dtype(np.complex64 if prec == "f32" else np.complex128)]
        g = torch.Generator(device="cuda").manual_seed(n % 1000 + batch)
        x = torch.empty(batch * n, dtype=cdt, device="cuda")
        torch.view_as_real(x).uniform_(-1, 1, generator=g)
        plan = commit(n, prec, batch, env)
        d0 = plan.info().dims[0]
        assert d0.tier == 3 and d0.n_factors == 3, (n, prec, d0.n_factors)
        if factors is not None:
            assert list(d0.factors[:3]) == factors, (n, prec, list(d0.factors[:3]))
        y = torch.empty_like(x)
        plan.compute_forward(x, y).wait()
        for b in sorted({0, batch - 1}):
            ref = np.fft.fft(x.view(batch, n)[b].cpu().numpy().astype(np.complex128))
            assert H.rel_l2(y.view(batch, n)[b].cpu().numpy(), ref) <= tol, (n, prec, batch, b)
            del ref
        z = torch.empty_like(x)
        plan.compute_backward(y, z).wait()
        err = float(((z / n - x).abs().double().pow(2).sum() / x.abs().double().pow(2).sum()).sqrt())
        assert err <= tol, (n, prec, batch, "round trip", err)
        del z
        y0 = torch.empty_like(x)
        e2 = dict(env or {})
        e2["PFFT_NO_THREE_STAGE"] = "1"
        try:
            plan2 = commit(n, prec, batch, e2)
        except pf.unsupported_configuration:
            plan2 = None
            assert n >= 1 << 27
        if plan2 is not None:
            assert plan2.info().dims[0].n_factors == 2
            plan2.compute_forward(x, y0).wait()
            diff = float(((y - y0).abs().double().pow(2).sum() / y0.abs().double().pow(2).sum()).sqrt())
            assert diff <= tol, (n, prec, batch, "vs the two-stage plan", diff)
        del y0, plan2
        # in place: S1 works on the user's buffer itself
        pin = commit(n, prec, batch, env, pf.placement.IN_PLACE)
        w = x.clone()
        pin.compute_forward(w).wait()
        assert torch.equal(w, y), (n, prec, batch, "in place")
        del w, x, y, plan, pin
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_three_stage_plan_split_storage():
    """The three-stage plan on SPLIT_COMPLEX data (runtime-specialised kernels on all three stages: planes -> planes with
    the store modifier in place on the output planes, planes -> interleaved scratch tiles, tiles -> planes): against NumPy,
    round trip, in place, and the two-stage plan of the same descriptor."""
    G, pf, torch = _mods()

    def commit(n, prec, batch, env=None, placement=None):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            d = G.make_descriptor([n], prec, batch=batch, storage=1)
            if placement is not None:
                d.placement = placement
            return d.commit()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v

    for n, prec, batch in ((1 << 23, "f32", 3), (5 << 20, "f64", 2), (3 << 22, "f32", 1)):
        rdt = torch.float32 if prec == "f32" else torch.float64
        tol = H.REL_L2_TOL[np.dtype(np.complex64 if prec == "f32" else np.complex128)]
        g = torch.Generator(device="cuda").manual_seed(n % 1000 + batch)
        re = torch.empty(batch * n, dtype=rdt, device="cuda").uniform_(-1, 1, generator=g)
        im = torch.empty(batch * n, dtype=rdt, device="cuda").uniform_(-1, 1, generator=g)
        plan = commit(n, prec, batch)
        assert plan.info().dims[0].tier == 3 and plan.info().dims[0].n_factors == 3, (n, prec)
        ore, oim = torch.empty_like(re), torch.empty_like(im)
        plan.compute_forward(re, im, ore, oim).wait()
        for b in sorted({0, batch - 1}):
            x = re.view(batch, n)[b].cpu().numpy().astype(np.float64) + 1j * im.view(batch, n)[b].cpu().numpy().astype(np.float64)
            ref = np.fft.fft(x)
            got = ore.view(batch, n)[b].cpu().numpy() + 1j * oim.view(batch, n)[b].cpu().numpy()
            assert H.rel_l2(got, ref) <= tol, (n, prec, batch, b)
            del x, ref, got
        zre, zim = torch.empty_like(re), torch.empty_like(im)
        plan.compute_backward(ore, oim, zre, zim).wait()
        num = ((zre / n - re).double().pow(2).sum() + (zim / n - im).double().pow(2).sum()).sqrt()
        den = (re.double().pow(2).sum() + im.double().pow(2).sum()).sqrt()
        assert float(num / den) <= tol, (n, prec, batch, "round trip")
        del zre, zim
        plan2 = commit(n, prec, batch, {"PFFT_NO_THREE_STAGE": "1"})
        assert plan2.info().dims[0].n_factors == 2
        o2re, o2im = torch.empty_like(re), torch.empty_like(im)
        plan2.compute_forward(re, im, o2re, o2im).wait()
        num = ((ore - o2re).double().pow(2).sum() + (oim - o2im).double().pow(2).sum()).sqrt()
        den = (o2re.double().pow(2).sum() + o2im.double().pow(2).sum()).sqrt()
        assert float(num / den) <= tol, (n, prec, batch, "vs the two-stage plan")
        del o2re, o2im, plan2
        pin = commit(n, prec, batch, None, pf.placement.IN_PLACE)
        wre, wim = re.clone(), im.clone()
        pin.compute_forward(wre, wim).wait()
        assert torch.equal(wre, ore) and torch.equal(wim, oim), (n, prec, batch, "in place")
        del wre, wim, re, im, ore, oim, plan, pin
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_four_step_half_pairs_and_split_choice():
    """Four-step lengths k * 2^m (plan_global.cpp, half pairs): the split takes a registered stage-B length (1024 / 512 / 256)
    as n2 and a SHORT runtime-specialised stage A of the same group width as n1 -- factors checked through the plan
    info --, the intermediate is group-major and stage B reads it in its tiled-input form (fp64: carrying the
    inter-stage twiddles on its loads).  Against NumPy on sampled transforms, round trip, and against the round's
    earlier plan of the same descriptor (PFFT_NO_HALF_PAIRS=1: balanced split, row-major intermediate); chunked
    batches with a ragged last chunk included."""
    G, pf, torch = _mods()

    def commit(n, prec, batch, env=None):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            return G.make_descriptor([n], prec, batch=batch).commit()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v

    cases = [  # (n, prec, batch, expected factors or None)
        (3 << 15, "f32", 5, [192, 512]), (5 << 15, "f32", 3, [160, 1024]), (3 << 16, "f32", 3, [192, 1024]),
        (7 << 14, "f32", 3, [224, 512]), (9 << 16, "f32", 2, [576, 1024]), (3 << 18, "f32", 2, [768, 1024]),
        (1 << 18, "f32", 3, [256, 1024]), (3 << 13, "f32", 7, [96, 256]), (5 << 18, "f32", 2, None),
        (3 << 20, "f32", 2, [1536, 2048]),
        # stage B n2 = 2048 reading tiles twice as wide as its groups (launch_tin_w) behind a 16-column stage A
        (1 << 21, "f32", 2, [1024, 2048]), (3 << 19, "f32", 2, [768, 2048]), (1 << 21, "f64", 2, [1024, 2048]),
        (12288, "f64", 5, None), (3 << 15, "f64", 3, None), (3 << 17, "f64", 2, [384, 1024]),
        (5 << 17, "f64", 2, [640, 1024]),
        (3 << 15, "f32", 700, None), (3 << 16, "f64", 170, None),  # chunked, ragged last chunk
    ]
    for n, prec, batch, factors in cases:
        cdt = torch.complex64 if prec == "f32" else torch.complex128
        tol = H.REL_L2_TOL[np.dtype(np.complex64 if prec == "f32" else np.complex128)]
        g = torch.Generator(device="cuda").manual_seed(n % 1000 + batch)
        x = torch.empty(batch * n, dtype=cdt, device="cuda")
        torch.view_as_real(x).uniform_(-1, 1, generator=g)
        # (24576 fp32 and 12288 fp64 fit the registers of one work-group since round 5 and would not take a four-step
        #  plan at all -- test_gpu_parity.py::test_register_resident_lengths has them; here they stay four-step cases)
        plan = commit(n, prec, batch, {"PFFT_NO_REGRES": "1"})
        d0 = plan.info().dims[0]
        assert d0.tier == 3, (n, prec)
        if factors is not None:
            assert list(d0.factors[:d0.n_factors]) == factors, (n, prec, list(d0.factors[:d0.n_factors]))
        y = torch.empty_like(x)
        plan.compute_forward(x, y).wait()
        for b in sorted({0, batch // 2, batch - 1}):
            ref = np.fft.fft(x.view(batch, n)[b].cpu().numpy().astype(np.complex128))
            assert H.rel_l2(y.view(batch, n)[b].cpu().numpy(), ref) <= tol, (n, prec, batch, b)
        z = torch.empty_like(x)
        plan.compute_backward(y, z).wait()
        err = float(((z / n - x).abs().double().pow(2).sum() / x.abs().double().pow(2).sum()).sqrt())
        assert err <= tol, (n, prec, batch, "round trip", err)
        y0 = torch.empty_like(x)
        commit(n, prec, batch, {"PFFT_NO_HALF_PAIRS": "1", "PFFT_NO_WIDE_TILES": "1",
                                "PFFT_NO_REGRES": "1"}).compute_forward(x, y0).wait()
        diff = float(((y - y0).abs().double().pow(2).sum() / y0.abs().double().pow(2).sum()).sqrt())
        assert diff <= tol, (n, prec, batch, "vs the plan without half pairs / wide tiles", diff)
        del x, y, z, y0, plan
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_runtime_specialised_stage_b_on_a_row_major_intermediate():
    """Four-step lengths without a registered stage pair and without a tiled intermediate (n2 or its first pass no multiple
    of stage A's group width: 10^6 = 1000 x 1000, 68640 = 104 x 660, 250000): both stages are compiled at commit, stage B
    reads the row-major intermediate f-fastest.  With PFFT_ROW_IN_MAX_N=0 no stage-B length is staged through LDS, so that
    form runs for every one of them (ragged passes included: 1000 points on 52 lanes x 2, 660 on 30 x 3 -- the latter on the
    register-resident stage kernel, stockham_strided_hx.hpp), stage A walks its groups XCD-contiguously where its row pitch
    is no multiple of a line.  Against NumPy, the round trip, and the default plan of the same descriptor.  (Round 5's opt-in
    row-lanes form of that stage B, PFFT_TIN_ROWS=1, was measured out and removed in round 6: profiles/r5_perf_tin_rows.txt.)"""
    G, pf, torch = _mods()

    def commit(n, prec, batch, env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            return G.make_descriptor([n], prec, batch=batch).commit()
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v

    for n, prec, batch in ((1000000, "f32", 3), (68640, "f32", 5), (68640, "f64", 3), (250000, "f32", 2), (1000000, "f64", 2)):
        cdt = torch.complex64 if prec == "f32" else torch.complex128
        tol = H.REL_L2_TOL[np.dtype(np.complex64 if prec == "f32" else np.complex128)]
        g = torch.Generator(device="cuda").manual_seed(n % 1000 + batch)
        x = torch.empty(batch * n, dtype=cdt, device="cuda")
        torch.view_as_real(x).uniform_(-1, 1, generator=g)
        plan = commit(n, prec, batch, {"PFFT_ROW_IN_MAX_N": "0"})
        assert plan.info().dims[0].tier == 3 and plan.info().knob_mask != 0, (n, prec)
        y = torch.empty_like(x)
        plan.compute_forward(x, y).wait()
        for b in sorted({0, batch - 1}):
            ref = np.fft.fft(x.view(batch, n)[b].cpu().numpy().astype(np.complex128))
            assert H.rel_l2(y.view(batch, n)[b].cpu().numpy(), ref) <= tol, (n, prec, batch, b)
        z = torch.empty_like(x)
        plan.compute_backward(y, z).wait()
        err = float(((z / n - x).abs().double().pow(2).sum() / x.abs().double().pow(2).sum()).sqrt())
        assert err <= tol, (n, prec, "round trip", err)
        y0 = torch.empty_like(x)
        commit(n, prec, batch, {}).compute_forward(x, y0).wait()
        diff = float(((y - y0).abs().double().pow(2).sum() / y0.abs().double().pow(2).sum()).sqrt())
        assert diff <= tol, (n, prec, "against the default plan", diff)
        del x, y, z, y0, plan
        torch.cuda.empty_cache()



def test_committed_descriptor_is_a_snapshot():
    """The reference copies `params` at commit (committed_descriptor_impl.hpp:716-725): changing the user's descriptor
    afterwards -- the common `d.number_of_transforms = ...; d.commit()` pattern -- must not change what an existing plan
    or its copies validate and run."""
    G, pf, torch = _mods()
    d = G.make_descriptor([256], "f32", batch=4)
    plan = d.commit()
    d.number_of_transforms = 64                      # the user's object moves on ...
    d.complex_storage = pf.complex_storage.SPLIT_COMPLEX
    big = d.commit()
    assert plan.params.number_of_transforms == 4 and big.params.number_of_transforms == 64
    x = torch.empty(4 * 256, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1)
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    plan.compute_forward(x, y).wait()                # ... the first plan still takes 4 interleaved transforms
    plan.copy().compute_forward(x, y2).wait()        # and so does its copy (re-derived nothing from `d`)
    ref = np.fft.fft(x.view(4, 256).cpu().numpy().astype(np.complex128), axis=1)
    assert H.rel_l2(y.view(4, 256).cpu().numpy(), ref) <= 2e-6 and torch.equal(y, y2)
    with pytest.raises(pf.invalid_configuration):    # too small for the 64-transform split plan
        big.compute_forward(x.real.contiguous(), x.imag.contiguous(), y.real.contiguous(), y.imag.contiguous())
