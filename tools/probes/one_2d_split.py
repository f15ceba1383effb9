"""one 2-D shape in SPLIT_COMPLEX (or interleaved) storage a few times, for rocprofv3 --kernel-trace --stats:
one_2d_split.py prec n0 n1 split|interleaved [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import portfft_amd as pf
prec, n0, n1, storage = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
batch = max(1, (1 << 30) // (n0 * n1 * (8 if prec == "f32" else 16)))
d = pf.descriptor([n0, n1], prec)
d.number_of_transforms = batch
rt = torch.float32 if prec == "f32" else torch.float64
ct = torch.complex64 if prec == "f32" else torch.complex128
if storage == "split":
    d.complex_storage = pf.complex_storage.SPLIT_COMPLEX
    args = [torch.empty(batch * n0 * n1, dtype=rt, device="cuda").uniform_(-1, 1) for _ in range(2)] + \
           [torch.empty(batch * n0 * n1, dtype=rt, device="cuda") for _ in range(2)]
else:
    x = torch.empty(batch * n0 * n1, dtype=ct, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1)
    args = [x, torch.empty_like(x)]
plan = d.commit()
plan.compute_forward(*args)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    plan.compute_forward(*args)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / reps
print("%s %dx%d %s: %.4f ms, %.3f of 8 TB/s" % (prec, n0, n1, storage, ms, 2.0 * batch * n0 * n1 * (8 if prec == "f32" else 16) / (ms * 1e-3) / 8e12))
