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

    def __init__(self, backend, device=None, timeout_s=120):
        import datetime
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.local_rank = env_world()
        self.device = device if device is not None else torch.device("cpu")
        self.active = self.world > 1
        self.backend = backend
        self.group = None  # the group the barrier / scalars travel on (None = the default gloo group)
        self.fallback_reason = None
        if not self.active:
            return
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
        if backend != "nccl":
            self.backend = "gloo"
            self.device = torch.device("cpu")
            return
        ok, reason, group = 1, "", None
        try:
            # collective: every rank calls new_group; the first all-reduce creates the RCCL communicator -- here, not
            # inside the timed region
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
            probe = torch.ones(1, device=self.device)
            dist.all_reduce(probe, group=group)
            torch.cuda.synchronize()
            if int(probe.item()) != self.world:
                ok, reason = 0, "probe all-reduce returned %r" % probe.item()
        except Exception as e:  # noqa: BLE001 -- reported below, decided collectively
            ok, reason = 0, "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # over gloo: every rank sees the same verdict
        if int(flag.item()) == 1:
            self.group = group
        else:
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
