"""Helpers for the -m gpu tests: run the product (HIP path through the C ABI) on numpy data via torch device
memory.  torch is only plumbing here (device buffers + stream)."""
import numpy as np

import helpers as H
import portfft_amd as pf


def torch_mod():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
    return torch


def make_descriptor(lengths, prec="f32", batch=1, storage=0, placement=1, fwd_strides=None, bwd_strides=None,
                    fwd_distance=None, bwd_distance=None, fwd_offset=0, bwd_offset=0, fwd_scale=1.0, bwd_scale=1.0):
    d = pf.descriptor(lengths, prec)
    d.number_of_transforms = batch
    d.complex_storage = pf.complex_storage(storage)
    d.placement = pf.placement(placement)
    if fwd_strides is not None:
        d.forward_strides = list(fwd_strides)
    if bwd_strides is not None:
        d.backward_strides = list(bwd_strides)
    if fwd_distance is not None:
        d.forward_distance = fwd_distance
    if bwd_distance is not None:
        d.backward_distance = bwd_distance
    d.forward_offset, d.backward_offset = fwd_offset, bwd_offset
    d.forward_scale, d.backward_scale = fwd_scale, bwd_scale
    return d


def run(desc, direction, in_buf, plan=None):
    """Execute on the GPU.  in_buf: flat complex numpy array laid out as the descriptor's input domain says.
    Returns the flat output buffer (numpy) of get_output_count elements; untouched elements keep the padding
    value, like the reference's tests (fft_test_utils.hpp:452)."""
    torch = torch_mod()
    plan = plan or desc.commit()
    n_out = desc.get_output_count(direction)
    split = desc.complex_storage == pf.complex_storage.SPLIT_COMPLEX
    in_place = desc.placement == pf.placement.IN_PLACE
    fn = plan.compute_forward if direction == pf.direction.FORWARD else plan.compute_backward
    if not split:
        x = torch.from_numpy(np.ascontiguousarray(in_buf)).cuda()
        if in_place:
            if x.numel() < n_out:
                x = torch.cat([x, torch.full((n_out - x.numel(),), H.PADDING_VALUE, dtype=x.dtype, device="cuda")])
            fn(x)
            y = x
        else:
            y = torch.full((n_out,), H.PADDING_VALUE, dtype=x.dtype, device="cuda")
            fn(x, y)
        plan.wait()
        return y.cpu().numpy()[:n_out]
    xr = torch.from_numpy(np.ascontiguousarray(in_buf.real)).cuda()
    xi = torch.from_numpy(np.ascontiguousarray(in_buf.imag)).cuda()
    if in_place:
        if xr.numel() < n_out:
            pad = torch.full((n_out - xr.numel(),), H.PADDING_VALUE, dtype=xr.dtype, device="cuda")
            xr, xi = torch.cat([xr, pad]), torch.cat([xi, pad])
        fn(xr, xi)
        yr, yi = xr, xi
    else:
        yr = torch.full((n_out,), H.PADDING_VALUE, dtype=xr.dtype, device="cuda")
        yi = torch.full((n_out,), H.PADDING_VALUE, dtype=xr.dtype, device="cuda")
        fn(xr, xi, yr, yi)
    plan.wait()
    return (yr.cpu().numpy() + 1j * yi.cpu().numpy())[:n_out].astype(in_buf.dtype)


def transform_packed(desc, direction, packed):
    """packed [batch, *dims] data of the input domain -> packed data of the output domain, through the
    descriptor's actual layout (scatter, run, gather)."""
    inv = pf.inv(direction)
    dims = desc.lengths
    b = desc.number_of_transforms
    buf = H.scatter(packed, desc.get_strides(direction), desc.get_distance(direction), desc.get_offset(direction),
                    desc.get_input_count(direction), pad=0.0 if desc.placement == pf.placement.IN_PLACE else H.PADDING_VALUE)
    out = run(desc, direction, buf)
    return H.gather(out, b, dims, desc.get_strides(inv), desc.get_distance(inv), desc.get_offset(inv)), out
