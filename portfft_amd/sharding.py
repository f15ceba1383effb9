"""Batch sharding across the GPUs of one node (BASELINE config 4).

The reference has no multi-device support (one sycl::queue per committed descriptor,
/root/reference/src/portfft/committed_descriptor_impl.hpp:108-111,716-725).  Batches never interact, so the path
shards as independent units: rank g of G owns the contiguous transforms [lo, hi) of the global batch, holds them in
its own HBM, and commits its own plan with number_of_transforms = hi - lo.  No FFT data crosses xGMI; the process
group (RCCL on GPUs, gloo on CPU in the tests) is used only for the barrier around the timed region and for the
max / gather of a few scalars.
"""
import os


def shard_range(total, world, rank):
    """Contiguous, balanced split of `total` transforms: the first (total % world) ranks get one extra."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad world/rank %r/%r" % (world, rank))
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def env_world():
    """(world_size, rank, local_rank) from the torch.distributed.run environment (1, 0, 0 when absent)."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


class process_group:
    """Thin wrapper: barrier, max-over-ranks and gather of scalars.

    The default group is always gloo (the control plane).  With backend 'nccl' (= RCCL on ROCm) an RCCL sub-group is
    created on top of it and probed with one all-reduce before any timing; whether it is used is decided
    COLLECTIVELY -- the ranks all-reduce (MIN) their success flags over gloo -- so either every rank runs its barrier
    and scalar collectives over RCCL or every rank runs them over gloo; a rank never changes backend on its own.
    `backend` after construction says which one carries the barrier ('nccl' or 'gloo')."""

    def __init__(self, backend, device=None, timeout_s=120, probe_timeout_s=45):
        import datetime
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.local_rank = env_world()
        self.device = device if device is not None else torch.device("cpu")
        # PFFT_BENCH_FORCE_DIST=1: build the control group and the RCCL sub-group even at world size 1, so that the
        # whole multi-rank code path (rendezvous, communicator, probe, barrier, max, gather) runs on a 1-GPU box
        force = os.environ.get("PFFT_BENCH_FORCE_DIST") == "1"
        self.active = self.world > 1 or force
        self.backend = backend
        self.group = None  # the group the barrier / scalars travel on (None = the default gloo group)
        self.fallback_reason = None
        if not self.active:
            return
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", str(self.rank))
            os.environ.setdefault("WORLD_SIZE", str(self.world))
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
        if backend != "nccl":
            self.backend = "gloo"
            self.device = torch.device("cpu")
            return
        # a failed or timed-out RCCL collective must RAISE in the caller (not abort the process from the watchdog
        # thread), or the collective verdict below is never reached
        os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")

        def agree(ok):  # over gloo: every rank sees the same verdict
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        reason, group = "", None
        # step 1: every rank creates the sub-group (no RCCL traffic yet) and the ranks agree that all of them could,
        # BEFORE the first RCCL collective -- a rank that failed here never leaves the others waiting inside RCCL
        try:
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=probe_timeout_s))
        except Exception as e:  # noqa: BLE001 -- reported below, decided collectively
            reason = "new_group: %s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
        usable = agree(group is not None)
        if usable:
            # step 2: the first all-reduce creates the RCCL communicator -- here, not inside the timed region; the
            # short group timeout bounds how long healthy ranks wait for one that failed inside RCCL
            ok = True
            try:
                probe = torch.ones(1, device=self.device)
                dist.all_reduce(probe, group=group)
                torch.cuda.synchronize()
                if int(probe.item()) != self.world:
                    ok, reason = False, "probe all-reduce returned %r" % probe.item()
            except Exception as e:  # noqa: BLE001
                ok, reason = False, "probe: %s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
            usable = agree(ok)
        if usable:
            self.group = group
        else:
            if group is not None:
                try:
                    dist.destroy_process_group(group)
                except Exception:  # noqa: BLE001 -- a half-built communicator may refuse; gloo carries on
                    pass
            self.backend = "gloo"
            self.device = torch.device("cpu")
            self.fallback_reason = reason or "RCCL failed on another rank"
            if self.rank == 0:
                import sys
                print("process_group: RCCL unusable on at least one rank (%s); every rank uses gloo for barrier/max"
                      % self.fallback_reason, file=sys.stderr)

    def barrier(self):
        if self.torch.cuda.is_available():
            self.torch.cuda.synchronize()
        if self.active:
            if self.group is not None:
                # an all-reduce on the RCCL group is the barrier (dist.barrier on NCCL groups needs device_ids)
                t = self.torch.zeros(1, device=self.device)
                self.dist.all_reduce(t, group=self.group)
            else:
                self.dist.barrier()
        if self.torch.cuda.is_available():
            self.torch.cuda.synchronize()

    def max(self, value):
        if not self.active:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def gather(self, values):
        """every rank contributes a list of floats; returns the [world][len] table on every rank"""
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.device)
        if not self.active:
            return [t.tolist()]
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [o.tolist() for o in out]

    def close(self):
        if self.active and self.dist.is_initialized():
            self.dist.destroy_process_group()
