"""Worker of tests/test_distributed_gloo.py: world_size-2 gloo run of the
multi-GPU plumbing on CPU.  Every rank takes its shard of a batch of
independent frames, "decodes" it with the oracle (no GPU here), and rank 0
checks that the gathered per-frame hashes equal the single-process result, that
the broadcast packed buffer arrived intact, and that max-over-ranks timing works."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from rawspeed_amd import abi, dist, synth
import golden_cases as G
from oracle_lib import HostImage, Oracle

N_FRAMES, W, H, BPS = 7, 256, 32, 14


def main():
    grp = dist.Group(backend="gloo", device=torch.device("cpu"))
    pitch = W * BPS // 8
    # rank 0 synthesises the whole batch and broadcasts the packed buffer
    if grp.rank == 0:
        frames = [synth.uniform(W * H, BPS, 100 + f).reshape(H, W) for f in range(N_FRAMES)]
        packed = np.concatenate([synth.pack_rows(px, BPS, abi.ORDER_MSB) for px in frames])
        buf = torch.from_numpy(packed.copy())
    else:
        buf = torch.zeros(N_FRAMES * H * pitch, dtype=torch.uint8)
    grp.broadcast_bytes(buf, src=0)
    packed = buf.numpy()
    lo, hi = dist.shard_range(N_FRAMES, grp.world, grp.rank)
    oracle = Oracle()
    mine = {}
    for f in range(lo, hi):
        d = abi.UnpackDesc(0, 0, W, H, pitch, BPS, abi.ORDER_MSB)
        img = HostImage(W, H)
        assert oracle.unpack(d, packed[f * H * pitch:(f + 1) * H * pitch], img) == 0
        mine[f] = G.image_hash(img.pixels())
    # The cfg-5 distribution code of bench.py (dist.distribute_units) with units that differ
    # in content AND in byte count: every rank must end up with exactly the bytes its own
    # plan expects (what it would have synthesised for its shard), in both modes.
    def unit(g):  # deterministic, size and content depend on g
        r = np.random.default_rng(7000 + g)
        return r.integers(0, 256, size=1000 + 137 * ((g * 5) % 7), dtype=np.uint8)
    n_units = 11
    want_mine = np.concatenate([unit(g) for g in range(*dist.shard_range(n_units, grp.world,
                                                                       grp.rank))] or
                               [np.zeros(0, np.uint8)])
    for mode in ("scatter", "broadcast"):
        calls = []

        def assemble(units):
            assert grp.rank == 0     # only rank 0 owns the batch
            calls.append(list(units))
            return torch.from_numpy(np.concatenate([unit(g) for g in units] or
                                                   [np.zeros(0, np.uint8)]))
        got, dt, moved = dist.distribute_units(
            grp, n_units, lambda g: unit(g).size, assemble, mode,
            lambda n: torch.zeros(n, dtype=torch.uint8))
        assert got.numpy().tobytes() == want_mine.tobytes(), (mode, grp.rank)
        total_b = sum(unit(g).size for g in range(n_units))
        own0 = sum(unit(g).size for g in range(*dist.shard_range(n_units, grp.world, 0)))
        assert moved == (total_b if mode == "broadcast" else total_b - own0), (mode, moved)
        assert dt >= 0.0
    # ... and the three-way record a default `bench.py --gpus N` puts in its line
    # (dist.distribute_all_modes): own shard, scatter, broadcast -- all three present, every
    # mode delivering the rank's own shard
    def assemble2(units):
        assert grp.rank == 0
        return torch.from_numpy(np.concatenate([unit(g) for g in units] or [np.zeros(0, np.uint8)]))
    recs, shards = dist.distribute_all_modes(
        grp, n_units, lambda g: unit(g).size, assemble2,
        lambda n: torch.zeros(n, dtype=torch.uint8), torch.from_numpy(want_mine.copy()),
        lambda a, b: a.numpy().tobytes() == b.numpy().tobytes())
    assert set(recs) == set(dist.MODES) == set(shards), recs
    for mode in dist.MODES:
        assert recs[mode]["delivers_the_ranks_own_shard"] is True, (mode, recs)
        assert shards[mode].numpy().tobytes() == want_mine.tobytes(), mode
    assert recs["own_shard"]["bytes"] == 0 and recs["broadcast"]["bytes"] > recs["scatter"]["bytes"] > 0
    grp.barrier()
    t_max = grp.max_over_ranks(0.5 + grp.rank)      # rank r "took" 0.5 + r seconds
    n_total = grp.sum_over_ranks(hi - lo)
    gathered = grp.gather_objects(mine, dst=0)
    if grp.rank == 0:
        merged = {}
        for g in gathered:
            assert not (set(g) & set(merged))            # shards are disjoint
            merged.update(g)
        want = {}
        for f in range(N_FRAMES):
            px = synth.uniform(W * H, BPS, 100 + f).reshape(H, W)
            want[f] = G.image_hash(px)
        ok = (merged == want and int(n_total) == N_FRAMES
              and abs(t_max - (0.5 + grp.world - 1)) < 1e-9)
        with open(sys.argv[1], "w") as fo:
            json.dump({"ok": bool(ok), "frames": len(merged), "t_max": t_max,
                       "world": grp.world}, fo)
    grp.close()


if __name__ == "__main__":
    main()
