"""one descriptor (manual-bench grammar of bench.py --manual) timed in a loop (for rocprofv3):
   one_desc.py <float|double> <key=value,...> [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from portfft_amd import manual_bench
import portfft_amd as pf
prec = "f64" if sys.argv[1] == "double" else "f32"
d = manual_bench.descriptor_from_string(sys.argv[2], prec)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
split = d.complex_storage == pf.complex_storage.SPLIT_COMPLEX
rt = torch.float32 if prec == "f32" else torch.float64
ct = torch.complex64 if prec == "f32" else torch.complex128
n_in, n_out = d.get_input_count(pf.direction.FORWARD), d.get_output_count(pf.direction.FORWARD)
if split:
    args = [torch.empty(n_in, dtype=rt, device="cuda").uniform_(-1, 1) for _ in range(2)] + [torch.empty(n_out, dtype=rt, device="cuda") for _ in range(2)]
else:
    x = torch.empty(n_in, dtype=ct, device="cuda"); torch.view_as_real(x).uniform_(-1, 1)
    args = [x, torch.empty(n_out, dtype=ct, device="cuda")]
if d.placement == pf.placement.IN_PLACE:
    args = args[:len(args) // 2]
plan = d.commit()
plan.compute_forward(*args); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps): plan.compute_forward(*args)
e.record(); torch.cuda.synchronize()
print("%s %s: %.4f ms" % (sys.argv[1], sys.argv[2], s.elapsed_time(e) / reps))
