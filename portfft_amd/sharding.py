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
    """Thin wrapper: barrier, max-over-ranks and gather of scalars.  backend 'nccl' (= RCCL on ROCm) or 'gloo'."""

    def __init__(self, backend, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.local_rank = env_world()
        self.device = device if device is not None else torch.device("cpu")
        self.active = self.world > 1
        self.backend = backend
        if self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend == "nccl":
                try:
                    dist.init_process_group("nccl", device_id=self.device)
                    # the first collective creates the RCCL communicator: fail here, not inside the timed region
                    probe = torch.zeros(1, device=self.device)
                    dist.all_reduce(probe)
                    torch.cuda.synchronize()
                except Exception as e:  # noqa: BLE001 -- RCCL unusable on this node: the scalars can travel over gloo
                    import sys
                    print("process_group: RCCL unavailable (%s); falling back to gloo for barrier/max" % e,
                          file=sys.stderr)
                    if dist.is_initialized():
                        dist.destroy_process_group()
                    dist.init_process_group("gloo")
                    self.backend = "gloo"
                    self.device = torch.device("cpu")
            else:
                dist.init_process_group(backend)

    def barrier(self):
        if self.torch.cuda.is_available():
            self.torch.cuda.synchronize()
        if self.active:
            self.dist.barrier()
        if self.torch.cuda.is_available():
            self.torch.cuda.synchronize()

    def max(self, value):
        if not self.active:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, values):
        """every rank contributes a list of floats; returns the [world][len] table on every rank"""
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.device)
        if not self.active:
            return [t.tolist()]
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.tolist() for o in out]

    def close(self):
        if self.active and self.dist.is_initialized():
            self.dist.destroy_process_group()
