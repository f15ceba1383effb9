"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol that
include/portfft_amd.h declares, and the descriptor logic (defaults, counts, layouts, validation, error types)
matches the reference's known answers and the oracle.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "portfft_amd.h")).read()
    declared = set(re.findall(r"\b(pfft_[a-z_]+)\s*\(", header))
    assert len(declared) >= 14
    lib = ctypes.CDLL(os.path.join(ROOT, "portfft_amd", "libportfft_amd.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export " + name
    from portfft_amd import _lib
    assert declared == set(_lib.SYMBOLS), "python binding out of sync with the header"


def test_struct_layout_matches_between_binding_and_oracle(oracle):
    from portfft_amd import _lib
    assert ctypes.sizeof(_lib.pfft_desc_t) == ctypes.sizeof(oracle.Desc)
    for (n1, _), (n2, _) in zip(_lib.pfft_desc_t._fields_, oracle.Desc._fields_):
        assert n1 == n2
        assert getattr(_lib.pfft_desc_t, n1).offset == getattr(oracle.Desc, n2).offset


def test_struct_layout_matches_the_header(tmp_path):
    """sizes and field offsets of the C structs in include/portfft_amd.h (compiled with gcc) against the ctypes mirror"""
    import subprocess
    from portfft_amd import _lib
    structs = {"pfft_desc_t": _lib.pfft_desc_t, "pfft_dim_info_t": _lib.pfft_dim_info_t,
               "pfft_plan_info_t": _lib.pfft_plan_info_t}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "portfft_amd.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for field, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, field, name, field))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for field, _ in cls._fields_:
            assert int(got["%s.%s" % (name, field)]) == getattr(cls, field).offset, (name, field)


def test_descriptor_defaults_and_getters():
    """test/unit_test/descriptor.cpp:29-74"""
    import portfft_amd as pf
    d = pf.descriptor([2, 3])
    assert d.lengths == [2, 3] and d.forward_strides == [3, 1] and d.backward_strides == [3, 1]
    assert d.forward_distance == 6 and d.backward_distance == 6 and d.number_of_transforms == 1
    assert d.forward_scale == 1 and d.backward_scale == 1
    assert d.complex_storage == pf.complex_storage.INTERLEAVED_COMPLEX and d.placement == pf.placement.OUT_OF_PLACE
    assert d.get_flattened_length() == 6
    d.forward_strides, d.backward_strides = [7, 2], [9, 3]
    assert d.get_strides(pf.direction.FORWARD) == [7, 2] and d.get_strides(pf.direction.BACKWARD) == [9, 3]
    d.forward_scale, d.backward_scale = 2.0, 3.0
    assert d.get_scale(pf.direction.FORWARD) == 2.0 and d.get_scale(pf.direction.BACKWARD) == 3.0
    assert pf.inv(pf.direction.FORWARD) == pf.direction.BACKWARD


def test_descriptor_buffer_count_kat():
    """test/unit_test/descriptor.cpp:76-109: 33 / 17"""
    import portfft_amd as pf
    d = pf.descriptor([2, 3])
    d.number_of_transforms = 2
    d.forward_strides, d.backward_strides = [8, 3], [2, 4]
    d.forward_distance, d.backward_distance = 15, 1
    d.forward_offset, d.backward_offset = 3, 5
    assert d.get_input_count(pf.direction.FORWARD) == 33 == d.get_output_count(pf.direction.BACKWARD)
    assert d.get_output_count(pf.direction.FORWARD) == 17 == d.get_input_count(pf.direction.BACKWARD)


def _pf_desc(case):
    import portfft_amd as pf
    name, lengths, fs, bs, fd, bd, batch, place = case
    d = pf.descriptor(lengths)
    d.number_of_transforms = batch
    d.placement = pf.placement(place)

    def dist(strides, dd):
        if dd is not None:
            return dd
        return int(np.prod([l * s for l, s in zip(lengths, strides)]))

    if fs is not None:
        d.forward_strides, d.forward_distance = list(fs), dist(fs, fd)
    if bs is not None:
        d.backward_strides, d.backward_distance = list(bs), dist(bs, bd)
    return d


@pytest.mark.parametrize("case", H.INVALID_CASES, ids=[c[0] for c in H.INVALID_CASES])
def test_invalid_descriptors_throw_invalid_configuration(case):
    """instantiate_fft_tests.hpp:322-373,406-411: EXPECT_THROW(desc.commit(queue), invalid_configuration);
    validation needs no device, and commit() runs it first."""
    import portfft_amd as pf
    d = _pf_desc(case)
    with pytest.raises(pf.invalid_configuration):
        d.validate()
    with pytest.raises(pf.invalid_configuration):
        d.commit()


def test_unsupported_configurations():
    import portfft_amd as pf
    with pytest.raises(pf.unsupported_configuration):
        pf.descriptor([64], "f32", pf.domain.REAL).validate()
    d = pf.descriptor([4, 4])
    d.number_of_transforms = 2
    d.forward_strides, d.forward_distance = [8, 2], 64
    with pytest.raises(pf.unsupported_configuration):
        d.validate()
    # a superset of the reference: UNPACKED layouts are not limited to subgroup-sized lengths here
    d = pf.descriptor([4096])
    d.forward_strides, d.forward_distance = [2], 8192
    d.validate()
    assert issubclass(pf.out_of_local_memory_error, pf.unsupported_configuration)
    assert issubclass(pf.invalid_configuration, pf.base_error)


def test_validation_counts_layouts_agree_with_oracle(oracle):
    """randomised descriptors: the product's host logic and the oracle's restatement of the reference agree"""
    import portfft_amd as pf
    rng = np.random.default_rng(7)
    L = oracle.lib()
    n_ok = n_bad = 0
    for _ in range(1500):
        rank = int(rng.integers(1, 4))
        lengths = [int(rng.integers(1, 9)) for _ in range(rank)]
        batch = int(rng.integers(1, 5))
        kind = rng.integers(0, 4)
        d = pf.descriptor(lengths)
        d.number_of_transforms = batch
        d.placement = pf.placement(int(rng.integers(0, 2)))
        if kind >= 1:
            d.forward_strides = [int(rng.integers(0, 20)) for _ in range(rank)]
            d.forward_distance = int(rng.integers(0, 40))
        if kind >= 2:
            d.backward_strides = [int(rng.integers(0, 20)) for _ in range(rank)]
            d.backward_distance = int(rng.integers(0, 40))
        if kind == 3 and rank == 1:
            d.forward_strides, d.forward_distance = [batch], 1
        od = oracle.make_desc(lengths, batch=batch, placement=int(d.placement), fwd_strides=d.forward_strides,
                              bwd_strides=d.backward_strides, fwd_distance=d.forward_distance,
                              bwd_distance=d.backward_distance)
        st, _ = oracle.validate(od)
        try:
            d.validate()
            got = oracle.OK
        except pf.invalid_configuration:
            got = oracle.INVALID
        except pf.unsupported_configuration:
            got = oracle.UNSUPPORTED
        assert got == st, (lengths, batch, d.forward_strides, d.backward_strides, d.forward_distance,
                           d.backward_distance, int(d.placement), got, st)
        n_ok += st == 0
        n_bad += st != 0
        if all(s > 0 for s in d.forward_strides + d.backward_strides):
            for dr in (0, 1):
                assert d.get_input_count(pf.direction(dr)) == L.pfo_input_count(od, dr)
                assert int(d.get_layout(pf.direction(dr))) == L.pfo_layout(od, dr)
    assert n_ok > 100 and n_bad > 100


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """the product has no CPU fallback: without the HIP library the import raises"""
    import importlib.util
    src = os.path.join(ROOT, "portfft_amd", "_lib.py")
    dst = tmp_path / "_lib.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("_lib_copy", str(dst))
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError):
        spec.loader.exec_module(mod)


def test_commit_without_gpu_reports_hip_error():
    import portfft_amd as pf
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(pf.hip_error):
        pf.descriptor([64]).commit()
