"""per-call cost of tiny transforms: back-to-back async executes, and a HIP-graph of 100 executes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import portfft_amd as pf
for name, dims, batch in (("C1 N=64 b=1", [64], 1), ("N=4096 b=1", [4096], 1), ("N=1200 b=1 (jit)", [1200], 1), ("16x16 b=1 (fused)", [16, 16], 1), ("N=65536 b=1 (four-step)", [65536], 1)):
    d = pf.descriptor(dims, "f32"); d.number_of_transforms = batch
    n = 1
    for l in dims: n *= l
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan = d.commit()
        x = torch.randn(n * batch, dtype=torch.complex64, device="cuda"); y = torch.empty_like(x)
        for _ in range(10): plan.compute_forward(x, y)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000): plan.compute_forward(x, y)
        t_issue = (time.perf_counter() - t0) / 2000
        s.synchronize()
        t_total = (time.perf_counter() - t0) / 2000
        t0 = time.perf_counter()
        for _ in range(2000): plan.compute_forward(x, y, want_event=False)
        s.synchronize()
        t_noev = (time.perf_counter() - t0) / 2000
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(100): plan.compute_forward(x, y)
        g.replay(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        s.synchronize()
        t_graph = (time.perf_counter() - t0) / 2000
    print("%-26s issue %.2f us/call, back-to-back %.2f us/call (without a completion event %.2f), in a graph of 100: %.2f us/call" % (name, t_issue * 1e6, t_total * 1e6, t_noev * 1e6, t_graph * 1e6))
