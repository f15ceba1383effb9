"""Scratch: throughput of the library across the configs of BASELINE.json and the reference's bench set."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import portfft_amd as pf

def run(name, lengths, batch, prec="f32", reps=8, **kw):
    n = 1
    for l in lengths: n *= l
    d = pf.descriptor(lengths, prec); d.number_of_transforms = batch
    for k, v in kw.items(): setattr(d, k, v)
    dt = torch.complex64 if prec == "f32" else torch.complex128
    cnt_in, cnt_out = d.get_input_count(pf.direction.FORWARD), d.get_output_count(pf.direction.FORWARD)
    split = int(d.complex_storage) == 1
    if split:
        rt = torch.float32 if prec == "f32" else torch.float64
        args = [torch.empty(cnt_in, dtype=rt, device="cuda").uniform_(-1, 1) for _ in range(2)] + [torch.empty(cnt_out, dtype=rt, device="cuda") for _ in range(2)]
    else:
        x = torch.empty(cnt_in, dtype=dt, device="cuda"); torch.view_as_real(x).uniform_(-1, 1)
        args = [x, torch.empty(cnt_out, dtype=dt, device="cuda")]
    plan = d.commit()
    plan.compute_forward(*args); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): plan.compute_forward(*args)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    esz = 8 if prec == "f32" else 16
    bytes_ = 2.0 * n * batch * esz
    fl = 5.0 * n * math.log2(n) * batch if n > 1 else 0
    info = plan.info()
    tiers = [info.dims[i].tier for i in range(len(lengths))]
    print("%-34s tiers=%s %9.4f ms  %6.2f TB/s (%4.1f%% of 8)  %7.1f GFLOP/s" % (name, tiers, ms, bytes_ / ms * 1e-9, bytes_ / ms * 1e-9 / 8 * 100, fl / ms * 1e-6), flush=True)

