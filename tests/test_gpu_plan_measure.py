"""Opt-in measured planning (PFFT_PLAN_MEASURE=1; plan_global.cpp measured_radices, jit.cpp spec_radix_candidates): the radix
sequence of a runtime-specialised packed length is timed at commit and recorded in the JIT cache directory.  The
reference's rule is static (src/portfft/committed_descriptor_impl.hpp:210-313); with the knob off nothing changes.
Every case runs in a process of its own: the kernel and choice tables are per process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import gpu_utils as G, helpers as H
n, batch = int(sys.argv[1]), 64
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
plan = G.make_descriptor([n], prec, batch=batch).commit()
d = plan.info().dims[0]
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.empty(batch * n, dtype=torch.complex64 if prec == "f32" else torch.complex128, device="cuda")
torch.view_as_real(x).uniform_(-1, 1, generator=g)
y = torch.empty_like(x)
plan.compute_forward(x, y).wait()
ref = np.fft.fft(x.view(batch, n)[5].cpu().numpy().astype(np.complex128))
print(json.dumps({"factors": [int(d.factors[i]) for i in range(d.n_factors)], "tier": int(d.tier),
                  "err": float(H.rel_l2(y.view(batch, n)[5].cpu().numpy(), ref))}))
""" % (ROOT, os.path.join(ROOT, "tests"))


def _commit(n, cache, measure, verbose=False, prec="f32"):
    env = dict(os.environ, PFFT_JIT_CACHE_DIR=str(cache))
    env.pop("PFFT_PLAN_MEASURE", None)
    if measure:
        env["PFFT_PLAN_MEASURE"] = "1"
    if verbose:
        env["PFFT_JIT_VERBOSE"] = "1"
    p = subprocess.run([sys.executable, "-c", CHILD, str(n), prec], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["err"] <= (2e-6 if prec == "f32" else 5e-15), out
    return out["factors"], p.stderr


def test_measured_choice_is_recorded_and_honoured(tmp_path):
    n = 6000
    static, _ = _commit(n, tmp_path / "a", measure=False)
    assert not list((tmp_path / "a").glob("choice_*")), "nothing is recorded with the knob off"
    measured, log = _commit(n, tmp_path / "b", measure=True, verbose=True)
    assert log.count("[portfft_amd plan] n=%d" % n) >= 2, "the candidates were timed: " + log[-400:]
    record = tmp_path / "b" / ("choice_gfx950_f32_%d.txt" % n)
    assert record.exists() and [int(v) for v in record.read_text().split()] == measured
    # another process, same cache directory: the record is read, nothing is timed, nothing is compiled
    again, log2 = _commit(n, tmp_path / "b", measure=True, verbose=True)
    assert again == measured and "[portfft_amd plan]" not in log2 and "[portfft_amd jit]" not in log2, log2[-400:]
    # the knob off: the static rule, whatever the directory holds
    off, _ = _commit(n, tmp_path / "b", measure=False)
    assert off == static
    # a hand-written record is honoured (this is the planner test of the cached choice on a device)
    d = tmp_path / "c"
    d.mkdir(mode=0o700)
    rec = d / ("choice_gfx950_f32_%d.txt" % n)
    rec.write_text("10 10 10 6\n")
    rec.chmod(0o600)
    forced, _ = _commit(n, d, measure=True)
    assert forced == [10, 10, 10, 6]
    rec.write_text("7 7 7\n")  # a record of another length is ignored
    ignored, _ = _commit(n, d, measure=False)
    assert ignored == static


def test_measured_four_step_split_is_recorded_and_honoured(tmp_path):
    """plan_global.cpp measured_split: the n1 x n2 of a GLOBAL-tier length without a registered stage pair"""
    n = 30000
    static, _ = _commit(n, tmp_path / "a", measure=False, prec="f64")
    assert static[0] * static[1] == n
    measured, log = _commit(n, tmp_path / "b", measure=True, verbose=True, prec="f64")
    assert log.count("[portfft_amd plan] n=%d split" % n) >= 4, "the candidates were timed: " + log[-400:]
    record = tmp_path / "b" / ("choice_split_gfx950_f64_%d.txt" % n)
    assert record.exists() and [int(v) for v in record.read_text().split()] == measured[:2]
    assert not (tmp_path / "b" / ("choice_gfx950_f64_%d.txt" % n)).exists(), "split records are kept apart from radix records"
    again, log2 = _commit(n, tmp_path / "b", measure=True, verbose=True, prec="f64")
    assert again == measured and "split" not in log2, log2[-400:]
    off, _ = _commit(n, tmp_path / "b", measure=False, prec="f64")
    assert off == static
    # a hand-written record is honoured; one whose product is another length is not
    d = tmp_path / "c"
    d.mkdir(mode=0o700)
    rec = d / ("choice_split_gfx950_f64_%d.txt" % n)
    rec.write_text("120 250\n")
    rec.chmod(0o600)
    forced, _ = _commit(n, d, measure=True, prec="f64")
    assert forced[:2] == [120, 250]
    rec.write_text("120 251\n")
    # (the record is ignored and the split is measured again -- and recorded over it)
    redone, _ = _commit(n, d, measure=True, prec="f64")
    assert redone[0] * redone[1] == n


def _tuned_entries():
    import re
    path = os.path.join(ROOT, "portfft_amd", "csrc", "tuned_gfx950.inc")
    out = []
    for ln in open(path):
        m = re.match(r"\{PFFT_PRECISION_(F32|F64), (\d+), (\d), (\d+), \{([\d, ]+)\}\}", ln.strip())
        if m:
            out.append(("f32" if m.group(1) == "F32" else "f64", int(m.group(2)), m.group(3) == "1",
                        [int(v) for v in m.group(5).split(",")]))
    return out


def test_the_tuned_table_is_used_by_default_and_can_be_turned_off(tmp_path):
    """portfft_amd/csrc/tuned_gfx950.inc (tools/gen_tuned_table.py): measured choices shipped with the library.  A sample
    of its entries: the default commit takes the entry's factors and computes the right answer; with
    PFFT_NO_TUNED_TABLE=1 the static rule is back."""
    entries = _tuned_entries()
    if not entries:
        pytest.skip("the table is empty")
    packed = [e for e in entries if not e[2]]
    splits = [e for e in entries if e[2]]
    sample = packed[:2] + packed[-1:] + splits[:1] + splits[-1:]
    # (four-step splits of lengths that fit the registers of one work-group -- fp32 to ~39 000 points, fp64 to ~15 800 --
    #  apply where the register-resident kernel is off or declines: PFFT_NO_REGRES=1 for every split entry)
    os.environ["PFFT_NO_REGRES"] = "1"
    try:
        _sampled_entries_are_taken(sample, tmp_path)
    finally:
        del os.environ["PFFT_NO_REGRES"]


def _sampled_entries_are_taken(sample, tmp_path):
    for prec, n, is_split, factors in sample:
        with_lanes = len(factors) >= 3 and factors[-2] == 0  # (a packed entry may end in "0, lanes")
        if with_lanes:
            factors = factors[:-2]
        got, _ = _commit(n, tmp_path / "t", measure=False, prec=prec)
        assert got[:len(factors)] == factors, (prec, n, got, factors)
        if with_lanes:
            continue  # (the static rule may take the same radices on other lanes)
        os.environ["PFFT_NO_TUNED_TABLE"] = "1"
        try:
            static, _ = _commit(n, tmp_path / "t", measure=False, prec=prec)
        finally:
            del os.environ["PFFT_NO_TUNED_TABLE"]
        assert static[:len(factors)] != factors, (prec, n, static)


ALL_TUNED_CHILD = r"""
import json, sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import gpu_utils as G, helpers as H
entries = json.loads(sys.argv[1])
bad, t0 = [], time.time()
import os
for prec, n, is_split, factors in entries:
    batch = max(2, min(64, (1 << 21) // n))
    # a four-step split of a length that the register-resident kernel takes by default is the plan of its A/B twin
    os.environ.pop("PFFT_NO_REGRES", None)
    if is_split:
        os.environ["PFFT_NO_REGRES"] = "1"
    plan = G.make_descriptor([n], prec, batch=batch).commit()
    d = plan.info().dims[0]
    got = [int(d.factors[i]) for i in range(d.n_factors)]
    want = factors[:-2] if (len(factors) >= 3 and factors[-2] == 0) else factors
    nbytes = n * (8 if prec == "f32" else 16)
    if got[:len(want)] != want and not is_split and 80 * 1024 < nbytes <= 152 * 1024:
        # an 80 ... 152 KiB length whose default plan is the two-per-CU register-resident one: the entry is the plan of the
        # LDS-resident twin (and the fallback when the pair's kernel needs scratch)
        os.environ["PFFT_NO_REGRES"] = "1"
        plan = G.make_descriptor([n], prec, batch=batch).commit()
        d = plan.info().dims[0]
        got = [int(d.factors[i]) for i in range(d.n_factors)]
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.empty(batch * n, dtype=torch.complex64 if prec == "f32" else torch.complex128, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1, generator=g)
    y = torch.empty_like(x)
    plan.compute_forward(x, y).wait()
    z = torch.empty_like(x)
    plan.compute_backward(y, z).wait()
    b = batch - 1
    ref = np.fft.fft(x.view(batch, n)[b].cpu().numpy().astype(np.complex128))
    err = float(H.rel_l2(y.view(batch, n)[b].cpu().numpy(), ref))
    rt = float(((z / n - x).abs().double().pow(2).sum() / x.abs().double().pow(2).sum()).sqrt())
    tol = 2e-6 if prec == "f32" else 5e-15
    if got[:len(want)] != want or not (err <= tol and rt <= tol):
        bad.append([prec, n, got, want, err, rt])
print(json.dumps({"checked": len(entries), "bad": bad, "seconds": time.time() - t0}))
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_every_tuned_entry_is_taken_and_computes_the_right_answer():
    """(Precision, length) entries of the shipped table (portfft_amd/csrc/tuned_gfx950.inc) -- radix sequences, lanes and
    four-step splits that replace the static rule BY DEFAULT -- committed with a small batch: the planner takes the
    entry's factors, the forward transform matches NumPy at the parity tolerance of tests/test_gpu_parity.py and the
    backward transform returns the input (the table was covered by 5 sampled entries and an out-of-suite
    A/B script)."""
    entries = _tuned_entries()
    if not entries:
        pytest.skip("the table is empty")
    # The suite's time budget (115 s of its 765 for this one test): a seeded sample of 48 entries, both ends of
    # the table and every precision / kind among them; PFFT_TEST_ALL_TUNED=1 (tools/probes/r6_full_checks.sh, its log under
    # profiles/) runs all of them -- to be done whenever the table or a kernel header changes.
    if os.environ.get("PFFT_TEST_ALL_TUNED", "0") in ("", "0") and len(entries) > 48:
        import random
        rng = random.Random(20261005)
        keep = {0, len(entries) - 1}
        for kind in sorted({(e[0], bool(e[2])) for e in entries}):
            keep.add(next(i for i, e in enumerate(entries) if (e[0], bool(e[2])) == kind))
        keep |= set(rng.sample(range(len(entries)), 48 - len(keep)))
        entries = [entries[i] for i in sorted(keep)]
    p = subprocess.run([sys.executable, "-c", ALL_TUNED_CHILD, json.dumps(entries)], capture_output=True, text=True,
                       timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    print("tuned table: %d entries in %.0f s" % (out["checked"], out["seconds"]))
    assert out["checked"] == len(entries) and not out["bad"], out["bad"][:10]
