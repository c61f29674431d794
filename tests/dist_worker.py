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
