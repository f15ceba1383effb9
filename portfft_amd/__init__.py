"""portfft_amd -- MI355X-native batched FFT engine behind portFFT's descriptor / committed_descriptor API.

Python mirror of the reference's plan-and-commit interface (src/portfft/descriptor.hpp,
src/portfft/committed_descriptor.hpp) on top of the C ABI of include/portfft_amd.h:

    import portfft_amd as pf
    desc = pf.descriptor([4096], "f32")
    desc.number_of_transforms = 65536
    plan = desc.commit()                  # validates, plans, uploads twiddles
    plan.compute_forward(x_gpu, y_gpu)    # torch tensors (or raw device pointers), asynchronous on the stream
"""
from .api import (  # noqa: F401
    base_error,
    committed_descriptor,
    event,
    complex_storage,
    descriptor,
    direction,
    domain,
    hip_error,
    internal_error,
    inv,
    invalid_configuration,
    layout,
    out_of_local_memory_error,
    placement,
    unsupported_configuration,
    version,
)

__all__ = [
    "descriptor", "committed_descriptor", "event", "domain", "complex_storage", "placement", "direction", "layout", "inv",
    "base_error", "internal_error", "invalid_configuration", "unsupported_configuration",
    "out_of_local_memory_error", "hip_error", "version",
]
