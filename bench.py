#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1] -- fp32 C2C 1-D forward, N=4096, batch=65536 per GPU,
out-of-place, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torch.distributed.run environment starts its own N ranks (a CHILD
torch.distributed.run process, before this process touches the GPU) and relays rank 0's line.

A step is one pass of the hot path (one compute_forward over the whole batch).  Batches shard over GPUs with no
data-path collective (weak scaling: every rank owns 65536 transforms); RCCL is used only for the barrier and the
max / gather of the per-rank elapsed times.  Rank 0 prints ONE JSON line.

Timing: the K timed steps run back to back between barrier + synchronize on both sides (wall clock -> `value`,
`ms_per_step`, nothing else in the loop); a separate loop of K steps (before it) brackets every launch with HIP events
on the plan's stream (-> `roofline.kernel_ms`, `roofline.achieved`), so the event records never sit in the timed region.

--config selects the other single-GPU workloads with the same JSON fields: c3 / c5 (BASELINE configs[2] / [4]) and
the reference's own bench set ref16 / ref256 / ref4096 / ref65536 (test/bench/portfft/bench_float.cpp:49-52), its
odd-composite regression sizes ref9800 / ref15360 / ref68640 and the four-step sizes g32_* / g64_*.
"""
import argparse
import glob
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("PFFT_JIT_CACHE_DIR", os.path.join(ROOT, "build", "jit_cache"))  # only non-headline configs compile
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "GFLOP/s (5Nlog2N) + achieved-HBM% for fp32 C2C 1D, 1/2/4/8 GPUs"

# name -> (lengths, transforms per GPU, precision, description, launches per execute)
WORKLOADS = {
    "c2": ([4096], 65536, "f32", "BASELINE configs[1]: fp32 C2C 1D forward N=4096 batch=65536 per GPU", 1),
    "c3": ([1 << 20], 128, "f64", "BASELINE configs[2]: fp64 C2C 1D N=1048576 batch=128", 2),
    "c5": ([1024, 1024], 256, "f32", "BASELINE configs[4]: fp32 C2C 2D 1024x1024 batch=256", 2),
    "ref16": ([16], 8 * 1024 * 1024, "f32", "reference bench set small_1d: fp32 N=16 batch=8Mi", 1),
    "ref256": ([256], 512 * 1024, "f32", "reference bench set medium_small_1d: fp32 N=256 batch=512Ki", 1),
    "ref4096": ([4096], 32 * 1024, "f32", "reference bench set medium_large_1d: fp32 N=4096 batch=32Ki", 1),
    "ref65536": ([65536], 2048, "f32", "reference bench set large_1d: fp32 N=65536 batch=2Ki", 2),
    # the reference's odd-composite GLOBAL-tier regression sizes (test/unit_test/instantiate_fft_tests.hpp:153-157),
    # 1 GiB per buffer
    "ref9800": ([9800], 13312, "f32", "reference GlobalTest regression size: fp32 N=9800 (2^3 5^2 7^2) batch=13Ki", 1),
    "ref15360": ([15360], 8704, "f32", "reference GlobalTest regression size: fp32 N=15360 (2^10 3 5) batch=8.5Ki", 1),
    "ref68640": ([68640], 1920, "f32", "reference GlobalTest regression size: fp32 N=68640 (2^5 3 5 11 13) batch=1920", 2),
    # the reference's GLOBAL-tier sizes (test/unit_test/instantiate_fft_tests.hpp:147-151) and the fp32 four-step sizes
    # beyond them, 1 GiB per buffer
    "g32_15": ([32768], 4096, "f32", "fp32 N=32768 batch=4Ki (reference GlobalTest size): register-resident work-group kernel", 1),
    "g32_14": ([16384], 8192, "f32", "fp32 N=16384 batch=8Ki: register-resident work-group kernel, two work-groups per CU", 1),
    "g64_13": ([8192], 8192, "f64", "fp64 N=8192 batch=8Ki (reference WorkgroupOrGlobal size): register-resident work-group kernel, two work-groups per CU", 1),
    "g64_14": ([16384], 4096, "f64", "fp64 N=16384 batch=4Ki (reference WorkgroupOrGlobal size): register-resident work-group kernel", 1),
    "g32_17": ([131072], 1024, "f32", "fp32 four-step N=131072 batch=1Ki (reference GlobalTest size)", 2),
    "g32_18": ([1 << 18], 512, "f32", "fp32 four-step N=2^18 batch=512", 2),
    "g32_19": ([1 << 19], 256, "f32", "fp32 four-step N=2^19 batch=256", 2),
    "g32_20": ([1 << 20], 128, "f32", "fp32 four-step N=2^20 batch=128", 2),
    "g32_21": ([1 << 21], 64, "f32", "fp32 four-step N=2^21 batch=64 (stage B reads tiles twice its group width)", 2),
    "g32_22": ([1 << 22], 32, "f32", "fp32 four-step N=2^22 batch=32", 2),
    "g32_24": ([1 << 24], 8, "f32", "fp32 three-stage N=2^24 batch=8", 3),
    "g64_16": ([65536], 1024, "f64", "fp64 four-step N=65536 batch=1Ki (reference GlobalTest size)", 2),
    "g64_17": ([131072], 512, "f64", "fp64 four-step N=131072 batch=512 (reference GlobalTest size)", 2),
    "g64_18": ([1 << 18], 256, "f64", "fp64 four-step N=2^18 batch=256", 2),
    # BATCH_INTERLEAVED on both sides (element i of transform b at i * batch + b; the reference's second first-class layout,
    # workgroup_dispatcher.hpp:148-229), 1 GiB per buffer: lengths whose full-width group is beyond the LDS -- one pass on the
    # one-per-CU register-resident strided kernel since round 6 (two column-shaped stages before)
    "bi32_2048": ([2048], 65536, "f32", "fp32 N=2048 batch=64Ki BATCH_INTERLEAVED in and out", 1),
    "bi64_2048": ([2048], 32768, "f64", "fp64 N=2048 batch=32Ki BATCH_INTERLEAVED in and out", 1),
}
BATCH_INTERLEAVED_WORKLOADS = ("bi32_2048", "bi64_2048")


def pmc_traffic(config):
    """HBM bytes per execute from the committed rocprofv3 PMC passes of the same command (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, separate passes: tools/summarize_pmc.py -> profiles/r*_pmc_traffic*.json); None when no summary is
    committed for this config.  PMC counters cannot be collected from inside this process, so the figure is reported
    with its provenance -- and with the planner's record of the launches it was measured on (`bench_kernel_label`,
    written next to the counters by tools/final_profiles_r3.sh): a plan that has since changed its kernels shows up as
    `traffic_matches_plan: false` in the line."""
    names = ["r*_pmc_traffic.json"] if config == "c2" else []
    names.append("r*_pmc_traffic_%s.json" % config)
    files = sorted(f for n in names for f in glob.glob(os.path.join(ROOT, "profiles", n)))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
            return float(d["traffic_bytes_per_launch"]), os.path.relpath(path, ROOT), d.get("bench_kernel_label")
        except (OSError, ValueError, KeyError):
            continue
    return None, None, None


def _numpy_slice_worker(args):
    import numpy as np
    seed, rows, n, seconds = args
    rng = np.random.Generator(np.random.SFC64(seed))
    x = (rng.uniform(-1, 1, (rows, n)) + 1j * rng.uniform(-1, 1, (rows, n))).astype(np.complex64)
    np.fft.fft(x)
    t0, reps = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        np.fft.fft(x)
        reps += 1
    return rows * reps, time.perf_counter() - t0


def cpu_baseline(n, sample_seconds=10.0):
    """CPU baselines on this host's cores, bounded samples of the headline workload (fp32 C2C forward, length n):
      value                  the oracle (oracle/: the reference's algorithm restated in C, kind 'port'), OpenMP over
                             transforms on every core.  It EMULATES the reference's sub-group lanes one by one (that is
                             what makes it a faithful restatement), which costs 15-20x against a tuned CPU FFT on the
                             same cores (the JSON note carries the ratio of the run): it is the parity checker timed,
                             not a statement about CPU FFT speed.
      numpy_allcore_gflops   NumPy's pocketfft on every core (a process pool over batch slices): what this host's
                             CPUs do on the same transforms with a production CPU library."""
    import multiprocessing as mp
    import numpy as np
    import oracle_binding as ob

    cores = max(1, min(os.cpu_count() or 1, ob.lib().pfo_max_threads()))
    rng = np.random.Generator(np.random.SFC64(0))
    flop = 5.0 * n * math.log2(n)

    def run(batch):
        x = (rng.uniform(-1, 1, (batch, n)) + 1j * rng.uniform(-1, 1, (batch, n))).astype(np.complex64).ravel()
        d = ob.make_desc([n], "f32", batch=batch)
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
    gflops = flop * batch * reps / total / 1e9
    # NumPy (pocketfft): one core, then every core through a process pool (each worker owns a slice of the batch)
    one = _numpy_slice_worker((1, 256, n, 1.0))
    numpy_1 = flop * one[0] / one[1] / 1e9
    workers = max(1, os.cpu_count() or 1)
    try:
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_numpy_slice_worker, [(10 + i, 256, n, 3.0) for i in range(workers)])
        numpy_all = sum(flop * rows / secs for rows, secs in parts) / 1e9
    except (OSError, ValueError):
        numpy_all, workers = None, 0
    return {"value": round(gflops, 3), "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "numpy_1core_gflops": round(numpy_1, 3),
            "numpy_allcore_gflops": None if numpy_all is None else round(numpy_all, 3), "numpy_cores": workers,
            "note": "oracle emulates the reference's sub-group lanes one by one (parity checker: %s as slow as pocketfft "
                    "on the same cores in this run); numpy_allcore_gflops is the credible CPU figure for this host"
                    % ("unknown" if not numpy_all else "%.0fx" % (numpy_all / max(gflops, 1e-9))),
            "sample": "oracle/ (reference algorithm restated in C, OpenMP over transforms), fp32 C2C forward N=%d, "
                      "batch=%d of the 65536 x %d passes, %.1f s; numpy: 256 transforms per worker, 3 s"
                      % (n, batch, reps, total)}


def kernel_label(plan, lengths):
    """name the launches of one execute from the planner's own record (pfft_plan_get_info)"""
    info = plan.info()
    tiers = {0: "register", 1: "workgroup", 2: "generic", 3: "global(four-step)"}
    parts = []
    for i in range(info.rank):
        d = info.dims[i]
        parts.append("dim%d n=%d %s radices/factors %s wg%d x%d ffts lds %d B" % (
            i, d.length, tiers.get(d.tier, "?"), "x".join(str(d.factors[k]) for k in range(d.n_factors)),
            d.workgroup_size, d.ffts_per_workgroup, d.lds_bytes))
    return "; ".join(parts)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torch.distributed.run environment: start the N ranks as a CHILD process
    (never exec: this process may not have touched the GPU yet, and must not be replaced either way) and relay.  A
    rendezvous port that is taken (another job on the node) is retried with a fresh child on another port."""
    base = int(os.environ.get("PFFT_BENCH_PORT", str(29500 + (os.getpid() % 400))))
    rc = 1
    for attempt in range(4):
        port = str(base + 997 * attempt)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__), "--gpus", str(args.gpus),
               "--steps", str(args.steps), "--warmup", str(args.warmup), "--config", args.config]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        if args.manual is not None:
            cmd += ["--manual", args.manual, "--precision", args.precision]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        sys.stderr.write(p.stderr)
        for ln in p.stdout.splitlines():
            if not ln.startswith("{"):
                print(ln, file=sys.stderr)
        if lines:
            print(lines[-1])
            return p.returncode
        rc = p.returncode if p.returncode != 0 else 1
        err = p.stderr.lower()
        if not ("address already in use" in err or "eaddrinuse" in err or "errno: 98" in err):
            break
        print("bench.py: rendezvous port %s is taken, retrying on another one" % port, file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="c2", choices=sorted(WORKLOADS),
                    help="c2 (default, the headline line); c3 / c5: the other single-GPU configs of BASELINE.json; "
                         "ref16 / ref256 / ref4096 / ref65536: the reference's own bench set; ref9800 / ref15360 / "
                         "ref68640: its odd-composite regression sizes; g32_* / g64_*: four-step (GLOBAL tier) sizes")
    ap.add_argument("--manual", metavar="KEY=VALUE,...",
                    help="any descriptor, in the grammar of the reference's bench_manual_float / bench_manual_double "
                         "(register_manual_bench.hpp), e.g. d=cpx,n=1024x1024,b=64,s=split,p=ip; overrides --config")
    ap.add_argument("--precision", default="float", choices=["float", "double"],
                    help="with --manual: bench_manual_float or bench_manual_double")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np

    manual_desc = None
    if args.manual is not None:
        from portfft_amd import manual_bench
        try:
            manual_desc = manual_bench.descriptor_from_string(args.manual, "f64" if args.precision == "double" else "f32")
        except manual_bench.bench_error as e:
            sys.exit("%s\n%s" % (e, manual_bench.help_text("bench.py")))
        args.config = "manual"
        lengths, batch_per_gpu, prec = manual_desc.lengths, manual_desc.number_of_transforms, manual_desc.scalar
        name, launches = "manual: %s:%s" % (args.precision, args.manual), None
    else:
        lengths, batch_per_gpu, prec, name, launches = WORKLOADS[args.config]
    n = int(np.prod(lengths))
    # the CPU baseline runs first, before this process initialises the GPU (its NumPy leg forks a worker pool)
    cpu = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.config == "c2" and not args.no_cpu_baseline:
        cpu = cpu_baseline(n)

    import torch
    import portfft_amd as pf
    from portfft_amd.sharding import env_world, process_group, shard_range

    cdt = torch.complex64 if prec == "f32" else torch.complex128
    esz = 8 if prec == "f32" else 16

    world, rank, local_rank = env_world()
    # PFFT_BENCH_FORCE_DIST=1: run the multi-rank plumbing (gloo control group + RCCL sub-group) at any world size
    distributed = world > 1 or os.environ.get("PFFT_BENCH_FORCE_DIST") == "1"
    if distributed:
        # one node: keep gloo's and RCCL's bootstrap sockets on the loopback interface (the container hostname may not
        # resolve to a usable interface); data never crosses these sockets -- the ranks exchange a few scalars
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    # one process per GPU; PFFT_BENCH_ONE_DEVICE=1 / PFFT_BENCH_BACKEND=gloo exist only to exercise the multi-rank
    # plumbing on a single-GPU box (every rank on device 0)
    one_device = os.environ.get("PFFT_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("PFFT_BENCH_BACKEND", "nccl")
    if distributed and not one_device and torch.cuda.device_count() < world:
        sys.exit("bench.py: rank %d of %d, but this process sees %d GPU(s) (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?): "
                 "one process per GPU needs WORLD_SIZE <= visible devices" % (rank, world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank if (distributed and not one_device) else 0)
    dev = torch.device("cuda", torch.cuda.current_device())
    pg = process_group(backend, dev)  # RCCL communicator created and probed here, before any timing
    if pg.fallback_reason:
        print("bench.py rank %d: control collectives fall back from RCCL: %s" % (rank, pg.fallback_reason), file=sys.stderr)
    # weak scaling: the global batch is batch_per_gpu * world transforms, sharded contiguously, no data-path collective
    lo, hi = shard_range(batch_per_gpu * world, world, rank)
    assert hi - lo == batch_per_gpu

    if manual_desc is not None:
        desc = manual_desc
    else:
        desc = pf.descriptor(lengths, prec)
        desc.number_of_transforms = batch_per_gpu
        if args.config in BATCH_INTERLEAVED_WORKLOADS:
            desc.forward_strides, desc.forward_distance = [batch_per_gpu], 1
            desc.backward_strides, desc.backward_distance = [batch_per_gpu], 1
    split = desc.complex_storage == pf.complex_storage.SPLIT_COMPLEX
    in_place = desc.placement == pf.placement.IN_PLACE
    n_in, n_out = desc.get_input_count(pf.direction.FORWARD), desc.get_output_count(pf.direction.FORWARD)
    if in_place:
        n_in = n_out = max(n_in, n_out)
    rdt = torch.float32 if prec == "f32" else torch.float64

    # synthetic inputs resident in HBM: uniform(-1, 1) real and imaginary parts, two buffers rotated per step
    # (split storage: two planes per buffer; an in-place descriptor transforms the rotated buffers themselves)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    def buffer(count, fill):
        if split:
            planes = [torch.empty(count, dtype=rdt, device=dev) for _ in range(2)]
            if fill:
                for p in planes:
                    p.uniform_(-1, 1, generator=gen)
            return planes
        x = torch.empty(count, dtype=cdt, device=dev)
        if fill:
            torch.view_as_real(x).uniform_(-1, 1, generator=gen)
        return [x]

    inputs = [buffer(n_in, True) for _ in range(2)]
    out = None if in_place else buffer(n_out, False)
    plan = desc.commit()  # torch's current stream

    def step(k):
        if in_place:
            plan.compute_forward(*inputs[k % 2], want_event=False)
        else:
            plan.compute_forward(*inputs[k % 2], *out, want_event=False)

    for w in range(args.warmup):
        step(w)

    # Order of the measurements behind the warm-up: copy probe, event-timed loop, wall-clock loop.  The driver runs this
    # with 5 warm-up steps (3.6 ms of GPU work) right after ~20 s of CPU baseline with the GPU idle; whatever loop came
    # first then ran 1.5-2.5 % slower than the one behind it (tools/wall_gap.py: wall clock, per-execute events, one
    # event pair around the loop and a HIP graph of the same K executes agree within 1 % once the device is warm), so
    # the loop that decides `value` runs last.

    # device-to-device copy of the same buffers (read + write = the same algorithmic bytes): the measured-bandwidth
    # yardstick SURVEY.md 8(d) asks for beside the nominal peak
    copy_ms = None
    if not in_place and not split and n_in == n_out:
        for _ in range(2):
            out[0].copy_(inputs[0][0])
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for k in range(10):
            out[0].copy_(inputs[k % 2][0])
        c1.record()
        torch.cuda.synchronize()
        copy_ms = c0.elapsed_time(c1) / 10

    # ---- per-launch device time (HIP events on the plan's stream = torch's current stream), K steps ----
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    for k in range(args.steps):
        starts[k].record()
        step(k)
        stops[k].record()
    torch.cuda.synchronize()
    kernel_ms = [s.elapsed_time(e) for s, e in zip(starts, stops)]
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)

    # ---- the timed region: exactly `steps` steps, wall clock, nothing but the launches inside ----
    pg.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0
    pg.barrier()
    elapsed = pg.max(my_elapsed)
    per_rank = pg.gather([my_elapsed])

    # parity spot check of the timed output (last step's input) against NumPy, through the descriptor's layout
    # (an in-place descriptor has consumed its inputs: one more execute on a fresh copy)
    def pick(buf, index):  # elements `index` of a buffer, as complex on the host
        if split:
            return (buf[0][index].cpu().numpy() + 1j * buf[1][index].cpu().numpy())
        return buf[0][index].cpu().numpy()

    if in_place:
        fresh = buffer(n_in, True)
        x_buf = [p.clone() for p in fresh]
        plan.compute_forward(*fresh, want_event=False)
        torch.cuda.synchronize()
        y_buf = fresh
    else:
        x_buf, y_buf = inputs[(args.steps - 1) % 2], out
    inv = pf.inv(pf.direction.FORWARD)

    def element_index(d):
        idx = np.zeros(lengths, dtype=np.int64)
        for axis, stride in enumerate(desc.get_strides(d)):
            shape = [1] * len(lengths)
            shape[axis] = lengths[axis]
            idx = idx + np.arange(lengths[axis], dtype=np.int64).reshape(shape) * stride
        return torch.from_numpy(idx).to(dev)

    idx_in, idx_out = element_index(pf.direction.FORWARD), element_index(inv)
    worst = 0.0
    for b in sorted({0, 777 % batch_per_gpu, (batch_per_gpu * 5) // 8, batch_per_gpu - 1}):
        xin = pick(x_buf, idx_in + (desc.get_offset(pf.direction.FORWARD) + b * desc.get_distance(pf.direction.FORWARD)))
        ref = desc.forward_scale * np.fft.fftn(xin.astype(np.complex128))
        got = pick(y_buf, idx_out + (desc.get_offset(inv) + b * desc.get_distance(inv)))
        worst = max(worst, float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
    assert worst < 1e-4, "parity check failed: rel-L2 %g" % worst

    straggler_exit = 0
    if rank == 0:
        flops_per_step = 5.0 * n * math.log2(n) * batch_per_gpu * world
        gflops = flops_per_step / (elapsed / args.steps) / 1e9
        alg_bytes = 2.0 * n * batch_per_gpu * esz  # per execute: every element read once + written once
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_label = pmc_traffic(args.config)
        live_label = kernel_label(plan, lengths)
        result = {
            "metric": METRIC if args.config == "c2" else "GFLOP/s (5Nlog2N) + achieved-HBM%% (%s)" % args.config,
            "value": round(gflops, 1),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": prec,
            "data": "synthetic",
            "config": {"workload": name + ", %s, %s, inputs resident in HBM" % (
                           "in-place" if in_place else "out-of-place", "split planes" if split else "interleaved"),
                       "lengths": lengths,
                       "batch_per_gpu": batch_per_gpu, "global_batch": batch_per_gpu * world,
                       "sharding": "batches, no data-path collective",
                       "barrier_backend": ("rccl" if pg.backend == "nccl" else pg.backend) if distributed else None,
                       "per_rank_elapsed_ms": [round(r[0] * 1e3, 3) for r in per_rank],
                       "parity_rel_l2_vs_numpy": worst},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_wall": round(alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_matches_plan": None if traffic_label is None else traffic_label == live_label,
                         "kernel": live_label, "hbm_passes_per_execute": launches,
                         "launches_per_execute": int(plan.info().launches[0]),
                         "kernel_ms": round(avg_kernel_ms, 5),
                         # the spread of the per-execute event times behind that mean (rocprofv3 sees 704-752 us on the
                         # headline kernel, profiles/r4_c2_kernel_stats.csv): [min, median, max]
                         "kernel_ms_min_median_max": [round(v, 5) for v in (min(kernel_ms), sorted(kernel_ms)[len(kernel_ms) // 2],
                                                                          max(kernel_ms))],
                         "copy_probe": None if copy_ms is None else {
                             "what": "torch device-to-device copy_ of the same input into the same output buffer",
                             "gbs": round(alg_bytes / (copy_ms * 1e-3) / 1e9, 1),
                             "frac_of_copy": round(achieved / (alg_bytes / (copy_ms * 1e-3) / 1e9), 4)},
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "achieved = algorithmic bytes of one execute / event-timed duration of its launches"
                                 + (" (two HBM passes: a plain copy pair of the same access shapes tops out at 0.41-0.44 "
                                    "of the peak, DESIGN.md section 5)" if launches == 2 else "")
                                 + (" (three HBM passes)" if launches == 3 else "")},
        }
        if pg.fallback_reason:
            result["config"]["rccl_fallback_reason"] = pg.fallback_reason
        if cpu is not None:
            result["cpu_baseline"] = cpu
        # `value` is taken on the slowest rank; a straggler would hide in that max: name it
        times = sorted(r[0] for r in per_rank)
        median = times[len(times) // 2]
        slow = [i for i, r in enumerate(per_rank) if r[0] > 1.15 * median]
        if world > 1 and slow:
            result["config"]["stragglers"] = {"ranks": slow, "median_ms": round(median * 1e3, 3)}
            print("bench.py: rank(s) %s took more than 1.15x the median step time (%.3f ms)" % (slow, median * 1e3),
                  file=sys.stderr)
            straggler_exit = 3 if os.environ.get("PFFT_BENCH_STRAGGLER_FATAL") == "1" else 0
        print(json.dumps(result))
    pg.close()
    if straggler_exit:
        sys.exit(straggler_exit)


if __name__ == "__main__":
    main()
