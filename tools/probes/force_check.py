import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import portfft_amd as pf
prec, n, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cdt = torch.complex64 if prec == "f32" else torch.complex128
x = torch.empty(batch * n, dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(-1, 1)
d = pf.descriptor([n], prec); d.number_of_transforms = batch
plan = d.commit()
y = torch.empty_like(x); plan.compute_forward(x, y).wait()
ref = np.fft.fft(x.view(batch, n)[0].cpu().numpy().astype(np.complex128))
got = y.view(batch, n)[0].cpu().numpy()
print(prec, n, os.environ.get("PFFT_JIT_STRIDED_FORCE"), "rel-L2", np.linalg.norm(got - ref) / np.linalg.norm(ref))
