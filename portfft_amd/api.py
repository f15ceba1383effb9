"""Host-side mirror of portfft::descriptor / portfft::committed_descriptor (names, fields, argument meaning and
error behaviour follow the reference so that tests read like the reference's own tests).

Reference: /root/reference/src/portfft/descriptor.hpp:43-271, committed_descriptor.hpp:58-310, enums.hpp:25-56,
common/exceptions.hpp:32-77.
"""
import ctypes as C
import enum

from . import _lib
from ._lib import lib


# ---- enums (enums.hpp:25-56) ----------------------------------------------------------------------------------
class domain(enum.IntEnum):
    REAL = 0
    COMPLEX = 1


class complex_storage(enum.IntEnum):
    INTERLEAVED_COMPLEX = 0
    SPLIT_COMPLEX = 1


class placement(enum.IntEnum):
    IN_PLACE = 0
    OUT_OF_PLACE = 1


class direction(enum.IntEnum):
    FORWARD = 0
    BACKWARD = 1


class layout(enum.IntEnum):
    PACKED = 0
    UNPACKED = 1
    BATCH_INTERLEAVED = 2


def inv(d):
    """enums.hpp:38: the opposite direction."""
    return direction.BACKWARD if d == direction.FORWARD else direction.FORWARD


# ---- exceptions (common/exceptions.hpp:32-77) -----------------------------------------------------------------
class base_error(RuntimeError):
    pass


class internal_error(base_error):
    pass


class invalid_configuration(base_error):
    pass


class unsupported_configuration(base_error):
    pass


class out_of_local_memory_error(unsupported_configuration):
    pass


class hip_error(base_error):
    pass


_STATUS_TO_EXC = {1: invalid_configuration, 2: unsupported_configuration, 3: out_of_local_memory_error,
                  4: internal_error, 5: hip_error}


def _check(status):
    if status != 0:
        raise _STATUS_TO_EXC.get(status, internal_error)(lib.pfft_last_error().decode())


def version():
    return lib.pfft_version().decode()


def _precision_code(p):
    s = str(p).lower()
    if p in (0, "f32") or "float32" in s or "complex64" in s or s in ("float", "single"):
        return 0
    if p in (1, "f64") or "float64" in s or "complex128" in s or s in ("double",):
        return 1
    raise invalid_configuration("unknown precision %r" % (p,))


class descriptor:
    """portfft::descriptor<Scalar, Domain> (descriptor.hpp:43-271): a plain parameter bag with the same fields."""

    def __init__(self, lengths, scalar="f32", dom=domain.COMPLEX):
        self.scalar = "f64" if _precision_code(scalar) else "f32"
        self.domain = domain(dom)
        self.lengths = [int(x) for x in lengths]
        self.forward_scale = 1.0
        self.backward_scale = 1.0
        self.number_of_transforms = 1
        self.complex_storage = complex_storage.INTERLEAVED_COMPLEX
        self.placement = placement.OUT_OF_PLACE
        # detail::get_default_strides (utils.hpp:190-201)
        strides, total = [0] * len(self.lengths), 1
        for i in reversed(range(len(self.lengths))):
            strides[i] = total
            total *= self.lengths[i]
        self.forward_strides = list(strides)
        self.backward_strides = list(strides)
        self.forward_distance = total
        self.backward_distance = total
        self.forward_offset = 0
        self.backward_offset = 0

    # -- C view ------------------------------------------------------------------------------------------------
    def _c(self):
        d = _lib.pfft_desc_t()
        if len(self.lengths) > _lib.MAX_RANK:
            raise unsupported_configuration("At most %d dimensions are supported" % _lib.MAX_RANK)
        d.precision = _precision_code(self.scalar)
        d.domain = int(self.domain)
        d.rank = len(self.lengths)
        d.complex_storage = int(self.complex_storage)
        d.placement = int(self.placement)
        d.n_forward_strides = len(self.forward_strides)
        d.n_backward_strides = len(self.backward_strides)
        for i, v in enumerate(self.lengths):
            d.lengths[i] = v
        for i, v in enumerate(self.forward_strides[:_lib.MAX_RANK]):
            d.forward_strides[i] = v
        for i, v in enumerate(self.backward_strides[:_lib.MAX_RANK]):
            d.backward_strides[i] = v
        d.forward_distance = self.forward_distance
        d.backward_distance = self.backward_distance
        d.forward_offset = self.forward_offset
        d.backward_offset = self.backward_offset
        d.number_of_transforms = self.number_of_transforms
        d.forward_scale = self.forward_scale
        d.backward_scale = self.backward_scale
        return d

    # -- getters (descriptor.hpp:161-260) --------------------------------------------------------------------
    def get_flattened_length(self):
        return int(lib.pfft_desc_flattened_length(C.byref(self._c())))

    def get_input_count(self, dir):
        return int(lib.pfft_desc_input_count(C.byref(self._c()), int(dir)))

    def get_output_count(self, dir):
        return int(lib.pfft_desc_output_count(C.byref(self._c()), int(dir)))

    def get_strides(self, dir):
        return self.forward_strides if dir == direction.FORWARD else self.backward_strides

    def get_distance(self, dir):
        return self.forward_distance if dir == direction.FORWARD else self.backward_distance

    def get_offset(self, dir):
        return self.forward_offset if dir == direction.FORWARD else self.backward_offset

    def get_scale(self, dir):
        return self.forward_scale if dir == direction.FORWARD else self.backward_scale

    def get_layout(self, dir):
        """detail::get_layout (utils.hpp:238-246)."""
        return layout(lib.pfft_desc_layout(C.byref(self._c()), int(dir)))

    def validate(self):
        """detail::validate::validate_descriptor (descriptor_validation.hpp:264-281); needs no device."""
        _check(lib.pfft_desc_validate(C.byref(self._c())))

    def commit(self, queue=None):
        """descriptor::commit(queue) (descriptor.hpp:152-156).  `queue` is a HIP stream: a torch.cuda.Stream, a raw
        hipStream_t value, or None for torch's current stream (the default stream without torch)."""
        return committed_descriptor(self, queue)


def _stream_handle(queue):
    if queue is None:
        try:
            import torch
            if torch.cuda.is_available():
                return int(torch.cuda.current_stream().cuda_stream)
        except ImportError:
            pass
        return 0
    if hasattr(queue, "cuda_stream"):
        return int(queue.cuda_stream)
    return int(queue)


def _ptr(x):
    """device pointer of a torch tensor / object with data_ptr() / raw integer"""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    return int(x)


def _is_complex(x):
    f = getattr(x, "is_complex", None)
    return bool(f()) if callable(f) else False


class committed_descriptor:
    """portfft::committed_descriptor<Scalar, Domain> (committed_descriptor.hpp:46-315)."""

    def __init__(self, desc, queue=None):
        self._plan = C.c_void_p()
        self.params = desc
        c = desc._c()
        _check(lib.pfft_plan_create(C.byref(c), C.c_void_p(_stream_handle(queue)), C.byref(self._plan)))

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            lib.pfft_plan_destroy(plan)

    def info(self):
        out = _lib.pfft_plan_info_t()
        _check(lib.pfft_plan_get_info(self._plan, C.byref(out)))
        return out

    def _compute(self, dir, args):
        n = len(args)
        if n == 1:  # in-place interleaved (committed_descriptor.hpp:171-176, 215-218)
            _check(lib.pfft_execute(self._plan, int(dir), _ptr(args[0]), _ptr(args[0])))
        elif n == 2 and self.params.complex_storage == complex_storage.SPLIT_COMPLEX and not _is_complex(args[0]):
            # in-place split (committed_descriptor.hpp:186-192, 228-232)
            _check(lib.pfft_execute_split(self._plan, int(dir), _ptr(args[0]), _ptr(args[1]), _ptr(args[0]),
                                          _ptr(args[1])))
        elif n == 2:  # out-of-place interleaved (committed_descriptor.hpp:242-246, 288-293)
            _check(lib.pfft_execute(self._plan, int(dir), _ptr(args[0]), _ptr(args[1])))
        elif n == 4:  # out-of-place split (committed_descriptor.hpp:258-263, 305-310)
            _check(lib.pfft_execute_split(self._plan, int(dir), _ptr(args[0]), _ptr(args[1]), _ptr(args[2]),
                                          _ptr(args[3])))
        else:
            raise invalid_configuration("compute_* takes (inout), (in, out), (inout_re, inout_im) or "
                                        "(in_re, in_im, out_re, out_im)")
        return self

    def compute_forward(self, *args):
        return self._compute(direction.FORWARD, args)

    def compute_backward(self, *args):
        return self._compute(direction.BACKWARD, args)

    def wait(self):
        """sycl::event::wait() of the event the reference returns."""
        _check(lib.pfft_plan_wait(self._plan))
