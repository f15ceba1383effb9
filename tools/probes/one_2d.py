"""one 2-D shape a few times (for rocprofv3 --kernel-trace --stats): one_2d.py prec n0 n1 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import portfft_amd as pf
prec, n0, n1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
batch = max(1, (1 << 30) // (n0 * n1 * (8 if prec == "f32" else 16)))
d = pf.descriptor([n0, n1], prec); d.number_of_transforms = batch
dt = torch.complex64 if prec == "f32" else torch.complex128
x = torch.empty(batch * n0 * n1, dtype=dt, device="cuda"); torch.view_as_real(x).uniform_(-1, 1)
y = torch.empty_like(x)
plan = d.commit()
for _ in range(reps):
    plan.compute_forward(x, y)
torch.cuda.synchronize()
