"""ctypes binding of oracle/libpfft_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libpfft_oracle.so")

MAX_RANK = 8
FORWARD, BACKWARD = 0, 1
WORKITEM, SUBGROUP, WORKGROUP, GLOBAL = 0, 1, 2, 3
PACKED, UNPACKED, BATCH_INTERLEAVED = 0, 1, 2
OK, INVALID, UNSUPPORTED, OUT_OF_LOCAL_MEMORY, INTERNAL, HIP_ERROR = range(6)

# the reference's default compile-time sub-group size (CMakeLists.txt:54) and a typical local memory size
DEFAULT_SG = 32
DEFAULT_LOCAL_MEM = 65536


class Desc(C.Structure):
    """pfft_desc_t of include/portfft_amd.h."""

    _fields_ = [
        ("precision", C.c_int32),
        ("domain", C.c_int32),
        ("rank", C.c_int32),
        ("complex_storage", C.c_int32),
        ("placement", C.c_int32),
        ("n_forward_strides", C.c_int32),
        ("n_backward_strides", C.c_int32),
        ("reserved_", C.c_int32),
        ("lengths", C.c_uint64 * MAX_RANK),
        ("forward_strides", C.c_uint64 * MAX_RANK),
        ("backward_strides", C.c_uint64 * MAX_RANK),
        ("forward_distance", C.c_uint64),
        ("backward_distance", C.c_uint64),
        ("forward_offset", C.c_uint64),
        ("backward_offset", C.c_uint64),
        ("number_of_transforms", C.c_uint64),
        ("forward_scale", C.c_double),
        ("backward_scale", C.c_double),
    ]


class Impl(C.Structure):
    _fields_ = [
        ("level", C.c_int32),
        ("n_kernels", C.c_int32),
        ("kernel_level", C.c_int32 * 32),
        ("kernel_length", C.c_int64 * 32),
        ("n_factors", C.c_int32 * 32),
        ("factors", (C.c_int32 * 4) * 32),
    ]


def make_desc(lengths, precision="f32", batch=1, storage=0, placement=1, fwd_strides=None, bwd_strides=None,
              fwd_distance=None, bwd_distance=None, fwd_offset=0, bwd_offset=0, fwd_scale=1.0, bwd_scale=1.0,
              domain=1):
    """Build a descriptor with the defaults of portfft::descriptor (descriptor.hpp:131-144)."""
    d = Desc()
    d.precision = 0 if precision in ("f32", np.float32, 0) else 1
    d.domain = domain
    d.rank = len(lengths)
    d.complex_storage = storage
    d.placement = placement
    total = 1
    default = [0] * len(lengths)
    for i in reversed(range(len(lengths))):
        default[i] = total
        total *= lengths[i]
    fs = list(default if fwd_strides is None else fwd_strides)
    bs = list(default if bwd_strides is None else bwd_strides)
    d.n_forward_strides, d.n_backward_strides = len(fs), len(bs)
    for i, v in enumerate(lengths):
        d.lengths[i] = v
    for i, v in enumerate(fs[:MAX_RANK]):
        d.forward_strides[i] = v
    for i, v in enumerate(bs[:MAX_RANK]):
        d.backward_strides[i] = v
    d.forward_distance = total if fwd_distance is None else fwd_distance
    d.backward_distance = total if bwd_distance is None else bwd_distance
    d.forward_offset, d.backward_offset = fwd_offset, bwd_offset
    d.number_of_transforms = batch
    d.forward_scale, d.backward_scale = fwd_scale, bwd_scale
    return d


def build():
    """(Re)build the oracle with its Makefile; a no-op when up to date."""
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.pfo_factorize.restype = C.c_int64
        L.pfo_factorize.argtypes = [C.c_int64]
        L.pfo_wi_temps.restype = C.c_int64
        L.pfo_wi_temps.argtypes = [C.c_int64]
        L.pfo_fits_in_wi.argtypes = [C.c_int64, C.c_int32]
        L.pfo_factorize_sg.restype = C.c_int64
        L.pfo_factorize_sg.argtypes = [C.c_int64, C.c_int32]
        L.pfo_fits_in_sg.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        L.pfo_prepare_implementation.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.POINTER(Impl),
                                                 C.c_char_p, C.c_size_t]
        L.pfo_validate.argtypes = [C.POINTER(Desc), C.c_int32, C.c_char_p, C.c_size_t]
        for f in (L.pfo_flattened_length,):
            f.restype = C.c_uint64
            f.argtypes = [C.POINTER(Desc)]
        for f in (L.pfo_input_count, L.pfo_output_count):
            f.restype = C.c_uint64
            f.argtypes = [C.POINTER(Desc), C.c_int32]
        L.pfo_layout.argtypes = [C.POINTER(Desc), C.c_int32]
        L.pfo_static_twiddle.restype = C.c_double
        L.pfo_static_twiddle.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.pfo_compute.argtypes = [C.POINTER(Desc), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_int64, C.c_int32, C.c_char_p, C.c_size_t]
        L.pfo_dft_1d.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def _check(st, buf):
    if st != OK:
        raise OracleError(st, buf.value.decode())


def validate(desc, sg=DEFAULT_SG):
    buf = C.create_string_buffer(512)
    st = lib().pfo_validate(C.byref(desc), sg, buf, 512)
    return st, buf.value.decode()


def prepare_implementation(n, scalar_bytes, sg=DEFAULT_SG, local_mem=DEFAULT_LOCAL_MEM):
    impl = Impl()
    buf = C.create_string_buffer(512)
    _check(lib().pfo_prepare_implementation(n, scalar_bytes, sg, local_mem, C.byref(impl), buf, 512), buf)
    return impl


def dft_1d(x, direction=FORWARD, level=-1, sg=DEFAULT_SG, local_mem=DEFAULT_LOCAL_MEM):
    x = np.ascontiguousarray(x)
    assert x.dtype in (np.complex64, np.complex128) and x.ndim == 1
    y = np.empty_like(x)
    buf = C.create_string_buffer(512)
    _check(lib().pfo_dft_1d(int(x.dtype == np.complex128), x.size, direction, level, sg, local_mem,
                            x.ctypes.data, y.ctypes.data, buf, 512), buf)
    return y


def compute(desc, direction, inp, out=None, inp_imag=None, out_imag=None, sg=DEFAULT_SG,
            local_mem=DEFAULT_LOCAL_MEM, threads=1):
    """Run the oracle on flat host arrays laid out as the descriptor says.  Returns `out`."""
    L = lib()
    inp = np.ascontiguousarray(inp)
    n_out = L.pfo_output_count(C.byref(desc), direction)
    if out is None:
        out = inp if desc.placement == 0 else np.zeros(n_out, dtype=inp.dtype)
    buf = C.create_string_buffer(512)
    st = L.pfo_compute(C.byref(desc), direction, inp.ctypes.data, out.ctypes.data,
                       None if inp_imag is None else inp_imag.ctypes.data,
                       None if out_imag is None else out_imag.ctypes.data, sg, local_mem, threads, buf, 512)
    _check(st, buf)
    return out
