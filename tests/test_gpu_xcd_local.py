"""The XCD-local single-launch four-step plan (portfft_amd/csrc/stockham_xcd.hpp, plan_global.cpp plan_xcd_local) through the
C ABI on an MI355X: fp32 N = 2^16, 2^17, 2^18, 2^20 and fp64 N = 2^16, 2^17, 2^18 (kernels_xcd.hip), the reference's
GLOBAL-tier size range
(ref: test/unit_test/instantiate_fft_tests.hpp:147-151, src/portfft/dispatcher/global_dispatcher.hpp:343-408).

Every plan is committed with PFFT_XCD_CHECK=1: the library then waits for each execute and raises when the persistent
launch needed its recovery launch (a hand-off wait gave up) -- on a healthy device that must not happen.  The recovery
itself is provoked and checked in test_a_launch_that_gives_up_is_recomputed_in_stream_order.  Results are compared with NumPy in double precision on sampled transforms (rel-L2 <= 2e-6, the bar of
tests/test_gpu_parity.py), with the two-launch plan of the same descriptor (PFFT_NO_XCD_LOCAL=1: another split of N, so
equal within the tolerance, not bit for bit -- the bit-for-bit comparison against the two launches of the SAME stage
kernels is tools/tune_xcd.hip), and through Parseval / round trips on every transform.
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

N = 1 << 18
TOL = 2e-6


def _mods():
    import gpu_utils as G
    import portfft_amd as pf
    return G, pf, G.torch_mod()


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(autouse=True)
def _checked_launches():
    with _env(PFFT_XCD_CHECK="1"):
        yield


def _random(torch, count, seed=5, prec="f32"):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.empty(count, dtype=torch.complex64 if prec == "f32" else torch.complex128, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1, generator=g)
    return x


def _check_samples(x, y, batch, samples, scale=1.0, n=N, tol=TOL):
    for b in samples:
        ref = np.fft.fft(x.view(batch, n)[b].cpu().numpy().astype(np.complex128)) * scale
        assert H.rel_l2(y.view(batch, n)[b].cpu().numpy(), ref) <= tol, ("transform", b)


# every registered pair (kernels_xcd.hip): (precision, log2 N, a batch above the plan's threshold with a ragged tail)
REGISTERED = [("f32", 16, 515), ("f32", 17, 301), ("f32", 18, 150), ("f32", 20, 67),
              ("f64", 16, 387), ("f64", 17, 259), ("f64", 18, 133)]


@pytest.mark.parametrize("prec,log2n,batch", REGISTERED)
def test_every_registered_pair_matches_numpy_and_the_two_launch_plan(prec, log2n, batch):
    """mixed stage configurations, groups side by side in one work-group, one to three work-groups per CU"""
    G, pf, torch = _mods()
    n = 1 << log2n
    tol = TOL if prec == "f32" else 5e-15
    with _env(PFFT_XCD_MIN_BATCH="64"):  # (below the measured crossover of the entry: correctness does not depend on it)
        plan = G.make_descriptor([n], prec, batch=batch).commit()
    assert list(plan.info().xcd_local) == [1, 1] and list(plan.info().launches) == [2, 2], "one launch + its recovery launch"
    x = _random(torch, batch * n, seed=log2n, prec=prec)
    y = torch.full_like(x, float("nan"))
    for _ in range(3):  # (repeated launches find the control block clean)
        plan.compute_forward(x, y).wait()
    _check_samples(x, y, batch, (0, batch // 2, batch - 1), n=n, tol=tol)
    ex = (x.view(batch, n).abs().double() ** 2).sum(dim=1)
    ey = (y.view(batch, n).abs().double() ** 2).sum(dim=1)
    assert float(((ey / (n * ex)) - 1).abs().max()) < (1e-5 if prec == "f32" else 1e-12)
    with _env(PFFT_NO_XCD_LOCAL="1"):
        plan2 = G.make_descriptor([n], prec, batch=batch).commit()
    assert list(plan2.info().xcd_local) == [0, 0] and min(plan2.info().launches) >= 2
    y2 = torch.empty_like(x)
    plan2.compute_forward(x, y2).wait()
    d = (y - y2).abs().double().pow(2).sum(dim=0).sqrt() / y2.abs().double().pow(2).sum(dim=0).sqrt()
    assert float(d) <= tol, float(d)
    del y2
    assert plan.info().xcd_recoveries == 0
    z = torch.empty_like(x)
    plan.compute_backward(y, z).wait()
    err = (z.view(batch, n) / n - x.view(batch, n)).abs().double().pow(2).sum(dim=1).sqrt() / ex.sqrt()
    assert float(err.max()) <= tol, float(err.max())


def test_xcd_local_plan_is_taken_and_matches_numpy_and_the_two_launch_plan():
    G, pf, torch = _mods()
    for batch in (256, 259, 515):  # the smallest batch that takes the plan (0.5 GiB of data), ragged counts
        for placement in (1, 0):
            desc = G.make_descriptor([N], "f32", batch=batch, placement=placement)
            plan = desc.commit()
            assert list(plan.info().xcd_local) == [1, 1]
            x = _random(torch, batch * N)
            y = x.clone() if placement == 0 else torch.full_like(x, float("nan"))
            (plan.compute_forward(y) if placement == 0 else plan.compute_forward(x, y)).wait()
            _check_samples(x, y, batch, (0, 1, batch // 2, batch - 2, batch - 1))
            # Parseval on every transform: no transform skipped, none computed from a stale intermediate
            ex = (x.view(batch, N).abs().double() ** 2).sum(dim=1)
            ey = (y.view(batch, N).abs().double() ** 2).sum(dim=1)
            assert float(((ey / (N * ex)) - 1).abs().max()) < 1e-5
            with _env(PFFT_NO_XCD_LOCAL="1"):
                plan2 = G.make_descriptor([N], "f32", batch=batch, placement=placement).commit()
            assert list(plan2.info().xcd_local) == [0, 0]
            y2 = x.clone() if placement == 0 else torch.empty_like(x)
            (plan2.compute_forward(y2) if placement == 0 else plan2.compute_forward(x, y2)).wait()
            d = (y - y2).abs().double().pow(2).sum(dim=0).sqrt() / y2.abs().double().pow(2).sum(dim=0).sqrt()
            assert float(d) <= TOL, (batch, placement, float(d))
            # backward through the same plan restores the input, on every transform
            z = torch.empty_like(x)
            plan.compute_backward(y, z).wait()
            err = (z.view(batch, N) / N - x.view(batch, N)).abs().double().pow(2).sum(dim=1).sqrt() / ex.sqrt()
            assert float(err.max()) <= TOL, (batch, placement, float(err.max()))
            del y, y2, z
        torch.cuda.empty_cache()


def test_small_batches_keep_the_two_launch_plan():
    G, pf, torch = _mods()
    def taken(lengths, prec, batch):
        return list(G.make_descriptor(lengths, prec, batch=batch).commit().info().xcd_local)
    assert taken([N], "f32", 16) == [0, 0]
    assert taken([N], "f32", 128) == [0, 0]  # (0.25 GiB of data)
    assert taken([1 << 16], "f32", 1024) == [0, 0]
    assert taken([1 << 16], "f32", 1536) == [1, 1]
    # other lengths have no registered pair
    assert taken([1 << 20], "f64", 256) == [0, 0]
    assert taken([1 << 15], "f32", 4096) == [0, 0]
    assert taken([3 << 16], "f32", 512) == [0, 0]
    assert taken([1 << 19], "f32", 512) == [0, 0] and taken([1 << 19], "f64", 256) == [0, 0]  # (de-registered in round 5)


def test_repeated_launches_offsets_scales_and_graph_replay():
    """The control block is left clean by every launch (the last work-group out clears it): 20 executes of one plan,
    then a captured execute replayed on new data.  Offsets and scales ride on the stage arguments."""
    G, pf, torch = _mods()
    batch = 200
    off_f, off_b = 24, 8
    desc = G.make_descriptor([N], "f32", batch=batch, fwd_offset=off_f, bwd_offset=off_b, fwd_scale=0.5, bwd_scale=2.0 / N)
    with _env(PFFT_XCD_MIN_BATCH="64"):
        plan = desc.commit()
    assert list(plan.info().xcd_local) == [1, 1]
    x = _random(torch, batch * N + off_f, seed=9)
    y = torch.full((batch * N + off_b,), 7.0, dtype=torch.complex64, device="cuda")
    for _ in range(20):
        plan.compute_forward(x, y)
    plan.wait()
    assert bool((y[:off_b] == 7.0).all()), "elements in front of the offset stay untouched"
    _check_samples(x[off_f:], y[off_b:], batch, (0, 77, batch - 1), scale=0.5)
    z = torch.zeros_like(x)
    plan.compute_backward(y, z).wait()  # backward: input at the backward offset, output at the forward offset
    err = float(((z[off_f:] - x[off_f:]).abs().double().pow(2).sum() / x[off_f:].abs().double().pow(2).sum()).sqrt())
    assert err <= TOL, err
    # graph capture: two kernel nodes (the launch and its recovery launch), no memset node -- replays leave the control
    # block as they found it
    s1 = torch.cuda.Stream()
    with _env(PFFT_XCD_MIN_BATCH="64"):
        plan_s = G.make_descriptor([N], "f32", batch=batch).commit(s1)
    xin = torch.zeros(batch * N, dtype=torch.complex64, device="cuda")
    out = torch.empty_like(xin)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):  # (the PFFT_XCD_CHECK wait is skipped inside a capture)
        plan_s.compute_forward(xin, out, want_event=False)
    torch.cuda.synchronize()
    for seed in (1, 2, 3):
        xr = _random(torch, batch * N, seed=seed)
        xin.copy_(xr)
        g.replay()
        torch.cuda.synchronize()
        _check_samples(xr, out, batch, (0, batch - 1))
    assert plan_s.info().xcd_recoveries == 0
    # and an ordinary checked launch of the replayed plan still finds its control block clean
    plan_s.compute_forward(xin, out).wait()
    _check_samples(xin, out, batch, (3,))


def test_copies_run_concurrently_with_partial_residency():
    """Two plans of one descriptor (each with its own slot rings and control block) on two streams at once: the work-groups of
    the two launches share the CUs, so neither launch has its full grid resident -- the hand-offs may not depend on that
    (every wait is for a task with a lower ticket, held by a running work-group)."""
    G, pf, torch = _mods()
    batch = 256
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with _env(PFFT_XCD_CHECK="0"):  # these plans enqueue without waiting, so that their launches overlap
        p1 = G.make_descriptor([N], "f32", batch=batch).commit(s1)
        p2 = G.make_descriptor([N], "f32", batch=batch).commit(s2)
        p3 = p1.copy()  # (a copy shares kernels and twiddles, owns its slot rings and control block; same stream as p1)
    x1, x2 = _random(torch, batch * N, seed=21), _random(torch, batch * N, seed=22)
    y1, y2 = torch.empty_like(x1), torch.empty_like(x2)
    torch.cuda.synchronize()
    for _ in range(5):
        p1.compute_forward(x1, y1)
        p2.compute_forward(x2, y2)
    torch.cuda.synchronize()
    p1.compute_forward(x1, y1).wait()
    p2.compute_forward(x2, y2).wait()
    assert p1.info().xcd_recoveries == 0 and p2.info().xcd_recoveries == 0, "no hand-off wait gave up in any launch"
    _check_samples(x1, y1, batch, (0, 100, batch - 1))
    _check_samples(x2, y2, batch, (0, 100, batch - 1))
    ex = (x1.view(batch, N).abs().double() ** 2).sum(dim=1)
    ey = (y1.view(batch, N).abs().double() ** 2).sum(dim=1)
    assert float(((ey / (N * ex)) - 1).abs().max()) < 1e-5
    y3 = torch.empty_like(x1)
    p3.compute_forward(x1, y3).wait()
    assert torch.equal(y3, y1), "a copy of the plan computes the same bits"


def _hip_event_synchronize(handle):
    """a raw hipEventSynchronize: what a caller does who waits on the returned event with HIP itself"""
    import ctypes
    for name in ("libamdhip64.so.7", "libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
            break
        except OSError:
            hip = None
    assert hip is not None, "HIP runtime not found"
    hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
    assert hip.hipEventSynchronize(ctypes.c_void_p(handle)) == 0


@pytest.mark.parametrize("prec,log2n,batch,iters", [("f32", 18, 256, 3), ("f32", 18, 259, 20), ("f32", 17, 520, 40),
                                                     ("f32", 20, 96, 20), ("f64", 16, 515, 30), ("f64", 18, 133, 8)])
def test_a_launch_that_gives_up_is_recomputed_in_stream_order(prec, log2n, batch, iters):
    """Every spin of the kernel is bounded; a launch that ran into a bound has skipped tasks.  PFFT_XCD_MAX_ITERS (read at
    commit) ends every work-group after that many tickets -- the bound exists against a runaway loop -- so part of the
    transforms are complete, part have their stage A done and part were never touched.  The recovery launch behind the
    persistent launch finishes the execute IN STREAM ORDER: a caller that waits on the returned event with HIP itself
    (no pfft_* call) reads the bits a healthy launch produces, out of place and in place, forward and backward, again and
    again (the recovery leaves the control block clean), and other plans of the process never notice
    (reference contract: a returned event means valid data, src/portfft/committed_descriptor.hpp:242-246)."""
    G, pf, torch = _mods()
    n = 1 << log2n
    with _env(PFFT_XCD_CHECK="0", PFFT_XCD_MIN_BATCH="64"):
        good = G.make_descriptor([n], prec, batch=batch).commit()
        with _env(PFFT_XCD_MAX_ITERS=str(iters)):
            bad = G.make_descriptor([n], prec, batch=batch).commit()
            bad_ip = G.make_descriptor([n], prec, batch=batch, placement=0).commit()
    assert list(good.info().xcd_local) == [1, 1] and list(bad.info().xcd_local) == [1, 1]
    x = _random(torch, batch * n, seed=31 + log2n, prec=prec)
    want = torch.empty_like(x)
    good.compute_forward(x, want).wait()
    _check_samples(x, want, batch, (0, batch - 1), n=n, tol=TOL if prec == "f32" else 5e-15)
    want_b = torch.empty_like(x)
    good.compute_backward(x, want_b).wait()
    for rep in range(2):
        y = torch.full_like(x, float("nan"))
        ev = bad.compute_forward(x, y)
        _hip_event_synchronize(ev.native)
        assert torch.equal(y, want), "out of place, forward"
        assert bad.info().xcd_recoveries == 2 * rep + 1
        ev = bad.compute_backward(x, y)
        _hip_event_synchronize(ev.native)
        assert torch.equal(y, want_b), "out of place, backward"
        assert bad.info().xcd_recoveries == 2 * rep + 2
        z = x.clone()
        ev = bad_ip.compute_forward(z)
        _hip_event_synchronize(ev.native)
        assert torch.equal(z, want), "in place: stage B again from the slot rings where the input is gone"
        assert bad_ip.info().xcd_recoveries == rep + 1
    # the healthy plan of the same process: unaffected, before and after
    y = torch.empty_like(x)
    good.compute_forward(x, y).wait()
    assert torch.equal(y, want) and good.info().xcd_recoveries == 0
    # a plan that gives up and recovers beside a healthy plan on another stream, launches interleaved without waiting:
    # both results right, the healthy plan's report untouched (no process-wide state)
    if log2n == 18 and iters == 20:
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with _env(PFFT_XCD_CHECK="0", PFFT_XCD_MIN_BATCH="64"):
            good2 = G.make_descriptor([n], prec, batch=batch).commit(s2)
            with _env(PFFT_XCD_MAX_ITERS=str(iters)):
                bad2 = G.make_descriptor([n], prec, batch=batch).commit(s1)
        y1, y2 = torch.empty_like(x), torch.empty_like(x)
        torch.cuda.synchronize()
        for _ in range(3):
            bad2.compute_forward(x, y1, want_event=False)
            good2.compute_forward(x, y2, want_event=False)
        torch.cuda.synchronize()
        assert torch.equal(y1, want) and torch.equal(y2, want)
        assert bad2.info().xcd_recoveries == 3 and good2.info().xcd_recoveries == 0
    # PFFT_XCD_CHECK=1 turns a recovery into an error (what the other tests of this file rely on)
    with _env(PFFT_XCD_CHECK="1", PFFT_XCD_MIN_BATCH="64", PFFT_XCD_MAX_ITERS=str(iters)):
        checked = G.make_descriptor([n], prec, batch=batch).commit()
    with pytest.raises(pf.internal_error, match="XCD-local"):
        checked.compute_forward(x, y)
    assert torch.equal(y, want), "... after the fact: the data is valid all the same"
