"""Scratch: how does the C2 kernel time depend on the relative placement of the input and output buffers?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import portfft_amd as pf

N, B = 4096, 65536
pool = torch.empty((24 << 30) // 8, dtype=torch.complex64, device="cuda")
torch.view_as_real(pool[: B * N]).uniform_(-1, 1)
d = pf.descriptor([N]); d.number_of_transforms = B
plan = d.commit()
base = pool.data_ptr()
print("pool base %x" % base)
def timeit(in_off, out_off, reps=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan.compute_forward(base + in_off, base + out_off)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s.record(); plan.compute_forward(base + in_off, base + out_off); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2], ts[0]
G2 = 2 << 30
M = 1 << 20
for in_off, delta in [(0, 0), (0, 256 * M), (0, 512 * M), (0, 768 * M), (0, 1024 * M), (0, 1536 * M), (0, 2048 * M), (0, 3072 * M), (0, 4096 * M),
                      (0, 6144 * M), (0, 8192 * M), (0, 12288 * M), (0, 16384 * M), (0, 20000 * M),
                      (4096 * M, -4096 * M - G2), (8192 * M, -8192 * M - G2), (1024 * M, 1024 * M), (1000 * M, 1000 * M), (1000 * M, 1300 * M)]:
    med, best = timeit(in_off, in_off + G2 + delta)
    print("in %6d MiB  out-in-2G %7d MiB  median %.4f ms (%.2f TB/s)  best %.4f ms" % (in_off // M, delta // M, med, 4.294967296 / med, best))
