"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) -- or gloo on CPU for the tests.

The decode path shards by independent units (frames / tiles) and needs NO
data-path collective; what is collective is (a) the barrier + max-over-ranks of
the timing, (b) gathering per-rank results, and (c) optionally distributing the
packed input from rank 0 (the only exchange the workload has: BASELINE config 5).
"""
import os


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(n_units, world, rank):
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one
    (cfg 5: 256 frames over 8 GPUs -> 32 each)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Group:
    def __init__(self, backend=None, device=None, force=False):
        """force: create the process group even for a world of one (tests of the
        collective path on a single-GPU box)."""
        import torch
        self.torch = torch
        self.world, self.rank, self.local_rank = env_world()
        self.enabled = self.world > 1 or force
        self.device = device
        if self.enabled:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "nccl", rank=self.rank,
                                        world_size=self.world, **kw)

    def barrier(self):
        if self.device is not None and self.device.type == "cuda":
            self.torch.cuda.synchronize()
        if self.enabled:
            self.dist.barrier()
            if self.device is not None and self.device.type == "cuda":
                self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if not self.enabled:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64,
                              device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if not self.enabled:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64,
                              device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def broadcast_bytes(self, tensor, src=0):
        """In-place broadcast of a uint8 tensor (the packed batch) from `src`."""
        if self.enabled:
            self.dist.broadcast(tensor, src=src)
        return tensor

    def scatter_shards(self, send, recv, src=0):
        """Every rank but `src` receives one shard into `recv`; `src` sends `send` (its own
        copy of a shard -- the shards of the synthetic batch are identical) to each of them
        as one group of point-to-point operations (ncclGroupStart/End underneath)."""
        if not self.enabled:
            return
        if self.rank == src:
            ops = [self.dist.P2POp(self.dist.isend, send, r)
                   for r in range(self.world) if r != src]
        else:
            ops = [self.dist.P2POp(self.dist.irecv, recv, src)]
        if not ops:  # a world of one
            return
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()

    def gather_objects(self, obj, dst=0):
        if not self.enabled:
            return [obj]
        out = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(obj, out, dst=dst)
        return out

    def close(self):
        if self.enabled and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
