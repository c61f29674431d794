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

    def scatter_shards(self, shards, recv, src=0):
        """Rank `src` holds `shards`, one uint8 tensor PER RANK (they differ in content and
        in size: shard r is what rank r's plan expects); every other rank receives its own
        into `recv`, which it has sized from the same unit sizes.  One group of
        point-to-point operations (ncclGroupStart/End underneath): 1/N of the bytes per
        xGMI link instead of the whole batch around a ring.  Returns the rank's shard."""
        if not self.enabled or self.world == 1:
            return shards[self.rank] if shards is not None else recv
        if self.rank == src:
            assert len(shards) == self.world
            ops = [self.dist.P2POp(self.dist.isend, shards[r], r)
                   for r in range(self.world) if r != src]
            mine = shards[src]
        else:
            ops = [self.dist.P2POp(self.dist.irecv, recv, src)]
            mine = recv
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()
        return mine

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


def distribute_units(grp, n_units, unit_bytes, assemble, mode, empty, sync=lambda: None,
                     clock=None):
    """Hand the packed input of a batch of `n_units` independent units (frames) out from
    rank 0 -- the only exchange the workload has (BASELINE configs[4]).

      unit_bytes(g)        packed size of global unit g (every rank can compute it)
      assemble(units)      rank 0 only: the packed bytes of the given global units, one tensor
      mode                 "broadcast": the WHOLE batch to everybody, a rank keeps its slice;
                           "scatter":   rank r is sent exactly ITS shard (shard_range), sized
                                        from unit_bytes -- the shards differ in content and size
      empty(nbytes)        an uninitialised uint8 tensor on the rank's device
      sync()               device synchronisation around the timed exchange (cuda)

    Returns (this rank's shard, seconds = max over ranks, bytes moved)."""
    import time
    clock = clock or time.perf_counter
    world, rank = grp.world, grp.rank
    sizes = [int(unit_bytes(g)) for g in range(n_units)]
    ranges = [shard_range(n_units, world, r) for r in range(world)]
    shard_bytes = [sum(sizes[lo:hi]) for lo, hi in ranges]
    lo, hi = ranges[rank]
    if mode == "broadcast":
        whole = assemble(range(n_units)) if rank == 0 else empty(sum(sizes))
        sync()
        grp.barrier()
        t0 = clock()
        grp.broadcast_bytes(whole, src=0)
        sync()
        grp.barrier()
        dt = grp.max_over_ranks(clock() - t0)
        start = sum(sizes[:lo])
        mine = whole[start:start + shard_bytes[rank]].clone()
        return mine, dt, int(whole.numel())
    if mode != "scatter":
        raise ValueError(mode)
    shards = [assemble(range(*ranges[r])) for r in range(world)] if rank == 0 else None
    recv = None if rank == 0 else empty(shard_bytes[rank])
    sync()
    grp.barrier()
    t0 = clock()
    mine = grp.scatter_shards(shards, recv, src=0)
    sync()
    grp.barrier()
    dt = grp.max_over_ranks(clock() - t0)
    return mine, dt, sum(shard_bytes) - shard_bytes[0]


MODES = ("own_shard", "scatter", "broadcast")


def distribute_all_modes(grp, n_units, unit_bytes, assemble, empty, own_shard, equal,
                         sync=lambda: None, clock=None, modes=MODES):
    """kernel-only scaling AND distribution-inclusive scaling in one run.  The
    packed batch reaches a rank's HBM in up to three ways --

      own_shard   the rank holds its shard already (synthesised / read by itself): no exchange
      scatter     rank 0 holds the batch and sends rank r exactly ITS shard (point to point)
      broadcast   rank 0 holds the batch and broadcasts all of it (BASELINE configs[4] as
                  BASELINE.json words it); a rank keeps its slice

    -- and every mode must leave the rank with the very bytes its plan expects, which
    `equal(got, own_shard)` checks on the rank's device.  Returns (records, shards):
    records[mode] = {"ms", "bytes", "gbps", "delivers_the_ranks_own_shard"} with ms the max
    over the ranks, shards[mode] the rank's packed input as that mode delivered it."""
    recs = {"own_shard": {"what": "every rank synthesises / holds its own shard: no exchange",
                          "ms": 0.0, "bytes": 0, "gbps": None,
                          "delivers_the_ranks_own_shard": True}}
    shards = {"own_shard": own_shard}
    for mode in modes:
        if mode == "own_shard":
            continue
        got, dt, moved = distribute_units(grp, n_units, unit_bytes, assemble, mode, empty,
                                          sync=sync, clock=clock)
        same = bool(equal(got, own_shard))
        all_same = grp.sum_over_ranks(1.0 if same else 0.0) == (grp.world if grp.enabled else 1)
        recs[mode] = {
            "what": ("grouped RCCL send/recv: rank 0 sends every other rank ITS shard (the "
                     "shards differ in content and size)") if mode == "scatter" else
                    "RCCL broadcast of the whole packed batch from rank 0; a rank keeps its slice",
            "ms": round(dt * 1e3, 3), "bytes": int(moved),
            "gbps": round(moved / max(dt, 1e-9) / 1e9, 2),
            "delivers_the_ranks_own_shard": bool(all_same)}
        shards[mode] = got
    return recs, shards
