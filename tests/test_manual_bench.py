"""Descriptor strings of the reference's manual benchmark (test/bench/portfft/register_manual_bench.hpp): grammar,
field mapping and error texts of portfft_amd.manual_bench (host logic, no GPU)."""
import pytest

import portfft_amd as pf
from portfft_amd import manual_bench as M


def test_fields_long_and_short_keys():
    d = M.descriptor_from_string("domain=complex,lengths=8x16,batch=3,fwd_strides=20x1,bwd_strides=1x8,fwd_dist=200,"
                                 "bwd_dist=128,scale=0.5,storage=split,placement=in_place", "f64")
    assert d.scalar == "f64" and d.domain == pf.domain.COMPLEX
    assert d.lengths == [8, 16] and d.number_of_transforms == 3
    assert d.forward_strides == [20, 1] and d.backward_strides == [1, 8]
    assert d.forward_distance == 200 and d.backward_distance == 128
    assert d.forward_scale == 0.5 and d.backward_scale == 0.5
    assert d.complex_storage == pf.complex_storage.SPLIT_COMPLEX and d.placement == pf.placement.IN_PLACE
    s = M.descriptor_from_string("d=cpx,n=4096,b=65536,s=int,p=oop")
    assert s.scalar == "f32" and s.lengths == [4096] and s.number_of_transforms == 65536
    assert s.forward_strides == [1] and s.forward_distance == 4096  # untouched fields keep the descriptor's defaults
    assert s.complex_storage == pf.complex_storage.INTERLEAVED_COMPLEX and s.placement == pf.placement.OUT_OF_PLACE
    for spelling, value in (("complex", 0), ("cpx", 0), ("interleaved", 0), ("int", 0), ("real_real", 1), ("rr", 1),
                            ("split", 1), ("sp", 1)):
        assert int(M.descriptor_from_string("d=cpx,n=4,s=" + spelling).complex_storage) == value
    assert M.descriptor_from_string("d=re,n=16").domain == pf.domain.REAL
    # the long spelling wins when both are given; an empty token ends the string
    assert M.descriptor_from_string("d=cpx,n=8,batch=5,b=7").number_of_transforms == 5
    assert M.descriptor_from_string("d=cpx,n=8,,b=7").number_of_transforms == 1


@pytest.mark.parametrize("text, message", [
    ("d=cpx,n", "Invalid token 'n'"),
    ("d=cpx,n=4,n=8", "Key can only be specified once: 'n'"),
    ("d=cpx,n=4,foo=1", "Invalid key: 'foo'"),
    ("d=cpx,n=", "Invalid 'n' value: ''"),
    ("n=4", "'domain' must be specified"),
    ("d=quaternion,n=4", "Invalid 'domain' value: 'quaternion'"),
    ("d=cpx", "'lengths' must be specified"),
    ("d=cpx,n=4x0", "Invalid 'lengths' value: '0' must be a positive integer"),
    ("d=cpx,n=4,b=-2", "Invalid 'batch' value: '-2' must be a positive integer"),
    ("d=cpx,n=4,b=many", "Invalid 'batch' value: 'many' must be a positive integer"),
    ("d=cpx,n=4,s=planar", "Invalid 'storage' value: 'planar'"),
    ("d=cpx,n=4,p=sideways", "Invalid 'placement' value: 'sideways'"),
])
def test_error_texts(text, message):
    with pytest.raises(M.bench_error) as e:
        M.descriptor_from_string(text)
    assert str(e.value) == message


def test_help_lists_every_key():
    h = M.help_text("bench.py")
    for long, short in M.ARG_KEYS:
        assert "'%s', '%s'" % (long, short) in h
