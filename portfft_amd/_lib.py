"""ctypes binding of libportfft_amd.so -- the C ABI declared in include/portfft_amd.h.

There is deliberately no fallback: if the HIP library has not been built this module raises ImportError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PORTFFT_AMD_LIBRARY selects another build of the same library (debug / experiment builds)
LIB_PATH = os.environ.get("PORTFFT_AMD_LIBRARY") or os.path.join(_HERE, "libportfft_amd.so")

MAX_RANK = 8
MAX_FACTORS = 16


class pfft_desc_t(C.Structure):
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


class pfft_dim_info_t(C.Structure):
    _fields_ = [
        ("length", C.c_uint64),
        ("tier", C.c_int32),
        ("n_factors", C.c_int32),
        ("factors", C.c_int32 * MAX_FACTORS),
        ("workgroup_size", C.c_int32),
        ("ffts_per_workgroup", C.c_int32),
        ("lds_bytes", C.c_uint64),
    ]


class pfft_plan_info_t(C.Structure):
    _fields_ = [
        ("rank", C.c_int32),
        ("n_compute_units", C.c_int32),
        ("twiddle_bytes", C.c_uint64),
        ("scratch_bytes", C.c_uint64),
        ("dims", pfft_dim_info_t * MAX_RANK),
        ("launches", C.c_int32 * 2),
        ("xcd_local", C.c_int32 * 2),
        ("xcd_recoveries", C.c_uint64),
        ("knob_mask", C.c_uint64),
    ]


# every symbol include/portfft_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "pfft_desc_init": (C.c_int, [C.POINTER(pfft_desc_t), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    "pfft_desc_validate": (C.c_int, [C.POINTER(pfft_desc_t)]),
    "pfft_desc_flattened_length": (C.c_uint64, [C.POINTER(pfft_desc_t)]),
    "pfft_desc_input_count": (C.c_uint64, [C.POINTER(pfft_desc_t), C.c_int32]),
    "pfft_desc_output_count": (C.c_uint64, [C.POINTER(pfft_desc_t), C.c_int32]),
    "pfft_desc_layout": (C.c_int32, [C.POINTER(pfft_desc_t), C.c_int32]),
    "pfft_plan_create": (C.c_int, [C.POINTER(pfft_desc_t), C.c_void_p, C.POINTER(C.c_void_p)]),
    "pfft_plan_destroy": (C.c_int, [C.c_void_p]),
    "pfft_plan_get_info": (C.c_int, [C.c_void_p, C.POINTER(pfft_plan_info_t)]),
    "pfft_execute": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "pfft_execute_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pfft_execute_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_void_p)]),
    "pfft_execute_split_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "pfft_event_wait": (C.c_int, [C.c_void_p]),
    "pfft_event_query": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pfft_event_destroy": (C.c_int, [C.c_void_p]),
    "pfft_queue_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_void_p)]),
    "pfft_queue_wait": (C.c_int, [C.c_void_p]),
    "pfft_plan_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pfft_plan_wait": (C.c_int, [C.c_void_p]),
    "pfft_last_error": (C.c_char_p, []),
    "pfft_status_string": (C.c_char_p, [C.c_int]),
    "pfft_version": (C.c_char_p, []),
}


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one process cannot both own the
    GPU, so when torch is installed it is imported first: libportfft_amd.so's DT_NEEDED libamdhip64.so.7 then
    resolves to the runtime torch already loaded (same SONAME) and streams / device pointers are shared."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def _load():
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "portfft_amd: %s is missing -- build it with `make -C portfft_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here means the library is stale
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()
