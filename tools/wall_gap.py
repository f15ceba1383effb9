"""Headline config (fp32 N=4096 x 65536): where do the ~18 us per step between the wall clock of K back-to-back
executes and the event-timed duration of one execute go?  
  (i)   K executes, wall clock (bench.py's `ms_per_step`) and per-execute events (bench.py's `kernel_ms`)
  (ii)  the same K executes captured once into a HIP graph, replayed: wall clock per step
  (iii) one pair of events around the whole K-loop (device time of the back-to-back sequence)
  (iv)  (i) with PFFT_UNIFORM_GRID=1 (run the script again with the variable set)
Usage: python tools/wall_gap.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import portfft_amd as pf

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n, batch = 4096, 65536
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    d = pf.descriptor([n], "f32"); d.number_of_transforms = batch
    plan = d.commit(s)
    xs = [torch.empty(n * batch, dtype=torch.complex64, device="cuda") for _ in range(2)]
    for x in xs: torch.view_as_real(x).uniform_(-1, 1)
    out = torch.empty_like(xs[0])
    def step(k): plan.compute_forward(xs[k % 2], out, want_event=False)
    for k in range(5): step(k)
    torch.cuda.synchronize()
    res = {}
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(K): step(k)
        torch.cuda.synchronize()
        res.setdefault("wall_ms_per_step", []).append((time.perf_counter() - t0) / K * 1e3)
        # per-execute events
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for k in range(K):
            ev[k][0].record(s); step(k); ev[k][1].record(s)
        torch.cuda.synchronize()
        res.setdefault("event_ms_per_execute", []).append(sum(a.elapsed_time(b) for a, b in ev) / K)
        # one pair of events around the whole loop
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for k in range(K): step(k)
        b.record(s); torch.cuda.synchronize()
        res.setdefault("event_ms_per_step_whole_loop", []).append(a.elapsed_time(b) / K)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for k in range(K): step(k)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        res.setdefault("graph_wall_ms_per_step", []).append((time.perf_counter() - t0) / K * 1e3)
    alg = 2.0 * n * batch * 8
    for k, v in res.items():
        m = sorted(v)[len(v) // 2]
        print("%-32s %s  median %.5f ms = %.4f of 8 TB/s" % (k, " ".join("%.5f" % x for x in v), m, alg / (m * 1e-3) / 8e12))
    print("uniform grid" if os.environ.get("PFFT_UNIFORM_GRID") == "1" else "two-tier grid", "steps", K)
