"""Host-side mirror of portfft::descriptor / portfft::committed_descriptor (names, fields, argument meaning and
error behaviour follow the reference so that tests read like the reference's own tests).

Reference: /root/reference/src/portfft/descriptor.hpp:43-271, committed_descriptor.hpp:58-310, enums.hpp:25-56,
common/exceptions.hpp:32-77.
"""
import copy as _copy
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


class event:
    """sycl::event of the reference: the completion of ONE compute_* submission (a hipEvent_t recorded behind its last
    kernel).  wait() blocks the host; pass it in another call's `dependencies` to order that call behind it."""

    def __init__(self, handle=None, plan=None):
        self._h = handle
        self._plan = plan  # keeps the plan alive; wait() on an event-less submission falls back to the stream

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.pfft_event_destroy(h)

    @property
    def native(self):
        return self._h

    def wait(self):
        if self._h:
            _check(lib.pfft_event_wait(self._h))
        elif self._plan is not None:
            self._plan.wait()
        return self

    def is_complete(self):
        if not self._h:
            return True
        done = C.c_int32(1)
        _check(lib.pfft_event_query(self._h, C.byref(done)))
        return bool(done.value)


def _dep_handle(d):
    """hipEvent_t of a dependency: an `event`, a torch.cuda.Event or a raw handle"""
    if isinstance(d, event):
        return d.native
    if hasattr(d, "cuda_event"):
        return int(d.cuda_event)
    return int(d) if d else None


class committed_descriptor:
    """portfft::committed_descriptor<Scalar, Domain> (committed_descriptor.hpp:46-315)."""

    def __init__(self, desc, queue=None, _clone_of=None):
        self._plan = C.c_void_p()
        if _clone_of is not None:
            # a copy shares the parent's snapshot: nothing is re-derived from a descriptor the user may have changed
            _check(lib.pfft_plan_clone(_clone_of._plan, C.byref(self._plan)))
            for name in ("params", "_device", "_torch", "_split", "_counts", "_scalar", "_real_dtype", "_cplx_dtype"):
                if hasattr(_clone_of, name):
                    setattr(self, name, getattr(_clone_of, name))
            self._no_deps = (C.c_void_p * 1)()
            return
        # the committed descriptor is a snapshot (the reference copies `params` at commit): the user's descriptor can be
        # modified and committed again without touching this plan
        self.params = _copy.deepcopy(desc)
        desc = self.params
        c = desc._c()
        _check(lib.pfft_plan_create(C.byref(c), C.c_void_p(_stream_handle(queue)), C.byref(self._plan)))
        self._device = None
        self._torch = None
        try:
            import torch
            self._torch = torch
            if torch.cuda.is_available():
                # the plan lives on the device of its queue: a torch stream knows its device, anything else was
                # committed on the current one
                dev = getattr(queue, "device", None)
                self._device = dev.index if dev is not None and getattr(dev, "index", None) is not None \
                    else torch.cuda.current_device()
        except ImportError:
            pass
        # element counts, storage and the dtypes a buffer may have are fixed here, so that a compute call costs a few
        # attribute reads
        self._split = desc.complex_storage == complex_storage.SPLIT_COMPLEX
        self._counts = {int(d): (desc.get_input_count(d), desc.get_output_count(d))
                        for d in (direction.FORWARD, direction.BACKWARD)}
        self._scalar = desc.scalar
        if self._torch is not None:
            f64 = desc.scalar == "f64"
            self._real_dtype = self._torch.float64 if f64 else self._torch.float32
            self._cplx_dtype = self._torch.complex128 if f64 else self._torch.complex64
        self._no_deps = (C.c_void_p * 1)()

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            lib.pfft_plan_destroy(plan)

    def copy(self):
        """the reference's copy constructor (committed_descriptor_impl.hpp:774-817): shares kernels and twiddles,
        owns its scratch"""
        return committed_descriptor(self.params, _clone_of=self)

    __copy__ = copy

    def info(self):
        out = _lib.pfft_plan_info_t()
        _check(lib.pfft_plan_get_info(self._plan, C.byref(out)))
        return out

    def _check_buffer(self, x, count, split_plane, what):
        """a torch tensor handed to compute_* must live on the plan's device, have the descriptor's element type, be
        contiguous and cover the descriptor's element count (raw pointers cannot be checked)"""
        if self._torch is None or not isinstance(x, self._torch.Tensor):
            return
        if not x.is_cuda:
            raise invalid_configuration("%s: the buffer is not in device memory" % what)
        if self._device is not None and x.device.index != self._device:
            raise invalid_configuration("%s: the buffer lives on device %s, the plan was committed on device %d"
                                        % (what, x.device.index, self._device))
        dt = x.dtype
        if split_plane:
            unit = 1
            ok = dt == self._real_dtype
        else:
            ok = dt == self._cplx_dtype or dt == self._real_dtype  # a real view counts two scalars per element
            unit = 2 if dt == self._real_dtype else 1
        if not ok:
            raise invalid_configuration("%s: dtype %s does not match the descriptor (%s %s storage)"
                                        % (what, dt, self._scalar, "split" if split_plane else "interleaved"))
        if not x.is_contiguous():
            raise invalid_configuration("%s: the buffer must be contiguous" % what)
        if x.numel() < count * unit:
            raise invalid_configuration("%s: %d elements, the descriptor addresses %d" % (what, x.numel() // unit, count))

    def _compute(self, dir, args, dependencies=None, want_event=True):
        n = len(args)
        split = self._split
        n_in, n_out = self._counts[int(dir)]
        if dependencies:
            deps = [h for h in (_dep_handle(d) for d in dependencies) if h]
            dep_arr = (C.c_void_p * max(len(deps), 1))(*deps)
            n_deps = len(deps)
        else:
            dep_arr, n_deps = self._no_deps, 0
        ev = C.c_void_p()
        ev_ref = C.byref(ev) if want_event else None
        if n == 1:  # in-place interleaved (committed_descriptor.hpp:171-176, 215-218)
            self._check_buffer(args[0], max(n_in, n_out), False, "inout")
            _check(lib.pfft_execute_ex(self._plan, int(dir), _ptr(args[0]), _ptr(args[0]), n_deps, dep_arr, ev_ref))
        elif n == 2 and split and not _is_complex(args[0]):
            # in-place split (committed_descriptor.hpp:186-192, 228-232)
            for a, w in zip(args, ("inout_real", "inout_imag")):
                self._check_buffer(a, max(n_in, n_out), True, w)
            _check(lib.pfft_execute_split_ex(self._plan, int(dir), _ptr(args[0]), _ptr(args[1]), _ptr(args[0]),
                                             _ptr(args[1]), n_deps, dep_arr, ev_ref))
        elif n == 2:  # out-of-place interleaved (committed_descriptor.hpp:242-246, 288-293)
            self._check_buffer(args[0], n_in, False, "in")
            self._check_buffer(args[1], n_out, False, "out")
            _check(lib.pfft_execute_ex(self._plan, int(dir), _ptr(args[0]), _ptr(args[1]), n_deps, dep_arr, ev_ref))
        elif n == 4:  # out-of-place split (committed_descriptor.hpp:258-263, 305-310)
            for a, w, c in zip(args, ("in_real", "in_imag", "out_real", "out_imag"), (n_in, n_in, n_out, n_out)):
                self._check_buffer(a, c, True, w)
            _check(lib.pfft_execute_split_ex(self._plan, int(dir), _ptr(args[0]), _ptr(args[1]), _ptr(args[2]),
                                             _ptr(args[3]), n_deps, dep_arr, ev_ref))
        else:
            raise invalid_configuration("compute_* takes (inout), (in, out), (inout_re, inout_im) or "
                                        "(in_re, in_im, out_re, out_im)")
        return event(ev.value if want_event else None, self)

    def compute_forward(self, *args, dependencies=None, want_event=True):
        """the USM overloads of committed_descriptor.hpp:171-310; `dependencies`: events (this module's, torch.cuda.Event
        or raw hipEvent_t) that must complete first; returns the event of this submission (want_event=False skips
        recording one: the returned object's wait() then waits for the plan's whole stream)."""
        return self._compute(direction.FORWARD, args, dependencies, want_event)

    def compute_backward(self, *args, dependencies=None, want_event=True):
        return self._compute(direction.BACKWARD, args, dependencies, want_event)

    def wait(self):
        """queue.wait(): everything submitted on the plan's stream has finished."""
        _check(lib.pfft_plan_wait(self._plan))
