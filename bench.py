#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1] -- fp32 C2C 1-D forward, N=4096, batch=65536 per GPU,
out-of-place, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (one compute_forward over the whole batch).  Batches shard over GPUs with no
data-path collective (weak scaling: every rank owns 65536 transforms); RCCL is used only for the barrier and the
max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("PFFT_JIT_CACHE_DIR", os.path.join(ROOT, "build", "jit_cache"))  # only --config c3/c5 paths compile
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

N = 4096
BATCH_PER_GPU = 65536
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, see tools/summarize_pmc.py and profiles/r*_pmc_traffic.json); None when no summary is committed.
    PMC counters cannot be collected from inside this process, so the figure comes from the same command run under
    rocprofv3 --pmc and is reported with its provenance."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return float(d["traffic_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        return None, None


def cpu_baseline(sample_seconds=12.0):
    """The oracle (CPU restatement of the reference algorithm, kind 'port') timed on this host's cores on a bounded
    sample of the same workload: fp32 N=4096 forward, batch sized to take about `sample_seconds`."""
    import numpy as np
    import oracle_binding as ob

    cores = max(1, min(os.cpu_count() or 1, ob.lib().pfo_max_threads()))
    rng = np.random.Generator(np.random.SFC64(0))

    def run(batch):
        x = (rng.uniform(-1, 1, (batch, N)) + 1j * rng.uniform(-1, 1, (batch, N))).astype(np.complex64).ravel()
        d = ob.make_desc([N], "f32", batch=batch)
        t0 = time.perf_counter()
        ob.compute(d, ob.FORWARD, x, threads=cores)
        return time.perf_counter() - t0

    probe = 16 * cores
    run(probe)  # warm-up (page faults, twiddle cache)
    t = run(probe)
    batch = int(max(probe, min(65536, probe * sample_seconds / max(t, 1e-6))))
    batch -= batch % cores
    reps, total = 0, 0.0
    while total < sample_seconds and reps < 64:  # many-core hosts finish the whole batch in ~2 s: repeat it
        total += run(batch)
        reps += 1
    gflops = 5.0 * N * math.log2(N) * batch * reps / total / 1e9
    # sanity row: NumPy's pocketfft on one core, same transform, ~1 s
    xs = (rng.uniform(-1, 1, (256, N)) + 1j * rng.uniform(-1, 1, (256, N))).astype(np.complex64)
    np.fft.fft(xs)
    t0, nrep = time.perf_counter(), 0
    while time.perf_counter() - t0 < 1.0:
        np.fft.fft(xs)
        nrep += 1
    numpy_gflops = 5.0 * N * math.log2(N) * 256 * nrep / (time.perf_counter() - t0) / 1e9
    return {"value": round(gflops, 3), "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "numpy_1core_gflops": round(numpy_gflops, 3),
            "sample": "oracle/ (reference algorithm restated in C, OpenMP over transforms), fp32 C2C forward N=%d, "
                      "batch=%d of the 65536 x %d passes, %.1f s" % (N, batch, reps, total)}


def other_config(args):
    """BASELINE configs[2] / configs[4] on one GPU: same protocol, same JSON fields (no cpu_baseline)."""
    import torch
    import portfft_amd as pf
    import numpy as np

    torch.cuda.set_device(0)
    if args.config == "c3":
        lengths, batch, prec, dt, name = [1 << 20], 128, "f64", torch.complex128, "BASELINE configs[2]: fp64 C2C 1D N=1048576 batch=128"
    else:
        lengths, batch, prec, dt, name = [1024, 1024], 256, "f32", torch.complex64, "BASELINE configs[4]: fp32 C2C 2D 1024x1024 batch=256"
    n = int(np.prod(lengths))
    d = pf.descriptor(lengths, prec)
    d.number_of_transforms = batch
    plan = d.commit()
    xs = []
    for _ in range(2):
        x = torch.empty(batch * n, dtype=dt, device="cuda")
        torch.view_as_real(x).uniform_(-1, 1)
        xs.append(x)
    y = torch.empty_like(xs[0])
    for w in range(args.warmup):
        plan.compute_forward(xs[w % 2], y)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for k in range(args.steps):
        plan.compute_forward(xs[k % 2], y)
    e.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = s.elapsed_time(e) / args.steps
    last = xs[(args.steps - 1) % 2].view([batch] + lengths)[batch - 1].cpu().numpy()
    ref = np.fft.fftn(last.astype(np.complex128))
    got = y.view([batch] + lengths)[batch - 1].cpu().numpy()
    err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    assert err < 1e-4, err
    esz = 16 if prec == "f64" else 8
    alg = 2.0 * n * batch * esz
    print(json.dumps({
        "metric": "GFLOP/s (5Nlog2N) + achieved-HBM%% (%s)" % args.config, "value": round(5.0 * n * math.log2(n) * batch / (elapsed / args.steps) / 1e9, 1),
        "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": prec, "data": "synthetic",
        "config": {"workload": name + ", out-of-place, inputs resident in HBM", "parity_rel_l2_vs_numpy": err},
        "roofline": {"bound": "hbm", "achieved": round(alg / (dev_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "two launches per execute (see DESIGN.md 3.3)", "kernel_ms": round(dev_ms, 5),
                     "algorithmic_bytes_per_launch": alg}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c5"],
                    help="c2 (default, the headline line): fp32 N=4096 batch=65536; c3: fp64 N=2^20 batch=128; "
                         "c5: fp32 2-D 1024x1024 batch=256 -- the other single-GPU configs of BASELINE.json, "
                         "reported with the same fields")
    args = ap.parse_args()
    if args.config != "c2":
        return other_config(args)

    import torch
    import portfft_amd as pf

    from portfft_amd.sharding import env_world, process_group, shard_range

    world, rank, local_rank = env_world()
    distributed = world > 1
    # one process per GPU; PFFT_BENCH_BACKEND=gloo + PFFT_BENCH_ONE_DEVICE=1 exist only to smoke-test the multi-rank
    # plumbing on a single-GPU box (every rank on device 0, scalar collectives over gloo)
    one_device = os.environ.get("PFFT_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("PFFT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank if (distributed and not one_device) else 0)
    cuda_dev = torch.device("cuda", torch.cuda.current_device())
    pg = process_group(backend, cuda_dev if backend == "nccl" else torch.device("cpu"))
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    # weak scaling: the global batch is BATCH_PER_GPU * world transforms, sharded contiguously
    lo, hi = shard_range(BATCH_PER_GPU * world, world, rank)
    assert hi - lo == BATCH_PER_GPU

    dev = torch.device("cuda", torch.cuda.current_device())
    # synthetic inputs resident in HBM: uniform(-1, 1) real and imaginary parts, two buffers rotated per step
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    inputs = []
    for _ in range(2):
        x = torch.empty(BATCH_PER_GPU * N, dtype=torch.complex64, device=dev)
        torch.view_as_real(x).uniform_(-1, 1, generator=gen)
        inputs.append(x)
    out = torch.empty(BATCH_PER_GPU * N, dtype=torch.complex64, device=dev)

    desc = pf.descriptor([N], "f32")
    desc.number_of_transforms = BATCH_PER_GPU
    plan = desc.commit()  # torch's current stream: the torch.cuda.Event timers below see the kernels

    barrier = pg.barrier

    for w in range(args.warmup):
        plan.compute_forward(inputs[w % 2], out)
    barrier()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        starts[k].record()
        plan.compute_forward(inputs[k % 2], out)
        stops[k].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = pg.max(elapsed)
    kernel_ms = [s.elapsed_time(e) for s, e in zip(starts, stops)]
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)

    # parity spot check of the timed output (last step's input) against NumPy on 4 transforms
    import numpy as np
    last_in = inputs[(args.steps - 1) % 2].view(BATCH_PER_GPU, N)
    worst = 0.0
    for b in (0, 777, 40000, BATCH_PER_GPU - 1):
        ref = np.fft.fft(last_in[b].cpu().numpy().astype(np.complex128))
        got = out.view(BATCH_PER_GPU, N)[b].cpu().numpy()
        worst = max(worst, float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
    assert worst < 1e-4, "parity check failed: rel-L2 %g" % worst

    if rank == 0:
        flops_per_step = 5.0 * N * math.log2(N) * BATCH_PER_GPU * world
        ms_per_step = elapsed / args.steps * 1e3
        gflops = flops_per_step / (elapsed / args.steps) / 1e9
        alg_bytes = 2.0 * N * BATCH_PER_GPU * 8  # per launch: every element read once + written once
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic()
        result = {
            "metric": "GFLOP/s (5Nlog2N) + achieved-HBM% for fp32 C2C 1D, 1/2/4/8 GPUs",
            "value": round(gflops, 1),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: fp32 C2C 1D forward N=4096 batch=65536 per GPU, out-of-place, "
                                   "interleaved, inputs resident in HBM", "n": N, "batch_per_gpu": BATCH_PER_GPU,
                       "global_batch": BATCH_PER_GPU * world, "sharding": "batches, no data-path collective",
                       "barrier_backend": ("rccl" if pg.backend == "nccl" else pg.backend) if distributed else None,
                       "parity_rel_l2_vs_numpy": worst},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "stockham_wg_prefetch_kernel<f32, 16x16x16, wg256, twiddles in VGPRs>", "kernel_ms": round(avg_kernel_ms, 5),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result))
    pg.close()


if __name__ == "__main__":
    main()
