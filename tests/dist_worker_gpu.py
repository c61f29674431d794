"""Worker of tests/test_gpu_dist_nccl.py: the multi-GPU plumbing of bench.py on real
devices -- backend "nccl" (= RCCL on ROCm), one process per visible GPU (1 is fine:
the same code path, collectives included).  Every rank decodes its shard of a batch
of independent LJPEG frames through the C-ABI; rank 0 distributes the packed batch
(broadcast, then grouped send/recv) and checks the gathered per-frame hashes against
the oracle's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from rawspeed_amd import abi, capi, dist
import cases
import golden_cases as G
from oracle_lib import HostImage, Oracle

N_FRAMES, W, H = 6, 512, 96


def make_batch():
    rng = np.random.default_rng(2024)
    descs, datas = [], []
    for f in range(N_FRAMES):
        d, data, _, _ = cases.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1,
                                              tile=(0, 0, W, H), mcu=(2, 1))
        descs.append(d)
        datas.append(np.ascontiguousarray(data))
    return descs, datas


def main():
    world, rank, local_rank = dist.env_world()
    torch.cuda.set_device(local_rank)
    grp = dist.Group(backend="nccl", device=torch.device("cuda", local_rank), force=True)
    assert grp.enabled and grp.dist.get_backend() == "nccl"
    descs, datas = make_batch()  # (deterministic: every rank builds the same descriptors)
    sizes = [d.size for d in datas]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    total = int(offs[-1])
    # rank 0 owns the packed batch; broadcast it, then also exercise the scatter path
    if rank == 0:
        buf = torch.from_numpy(np.concatenate(datas)).cuda()
    else:
        buf = torch.zeros(total, dtype=torch.uint8, device="cuda")
    grp.broadcast_bytes(buf, src=0)
    lo, hi = dist.shard_range(N_FRAMES, grp.world, grp.rank)
    # ... and the scatter path: rank r gets exactly ITS shard (the frames differ in size)
    dev = [torch.from_numpy(d).cuda() for d in datas]
    mine_sc, _, _ = dist.distribute_units(
        grp, N_FRAMES, lambda g: sizes[g],
        lambda units: torch.cat([dev[g] for g in units]) if len(units) else
        torch.empty(0, dtype=torch.uint8, device="cuda"), "scatter",
        lambda n: torch.zeros(n, dtype=torch.uint8, device="cuda"),
        sync=torch.cuda.synchronize)
    assert torch.equal(mine_sc, buf[int(offs[lo]):int(offs[hi])])
    ctx = capi.Context(local_rank)
    op = (W * 2 + 15) // 16 * 16
    jobs = []
    for k, f in enumerate(range(lo, hi)):
        j = abi.LJpegJob()
        j.desc = descs[f]
        j.in_offset, j.in_bytes = int(offs[f]), sizes[f]
        j.img_offset = k * op * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = op, W, H, 1, 1
        jobs.append(j)
    mine = {}
    if jobs:
        out = torch.zeros(len(jobs) * op * H, dtype=torch.uint8, device="cuda")
        plan = ctx.ljpeg_plan(jobs)
        plan.run(buf.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        rc, st, _ = plan.results()
        assert rc == 0, (rc, st)
        for k, f in enumerate(range(lo, hi)):
            px = out[k * op * H:(k + 1) * op * H].cpu().numpy().view(np.uint16) \
                .reshape(H, op // 2)[:, :W]
            mine[f] = G.image_hash(px)
    grp.barrier()
    t_max = grp.max_over_ranks(1.0 + grp.rank)
    n_total = grp.sum_over_ranks(hi - lo)
    gathered = grp.gather_objects(mine, dst=0)
    if rank == 0:
        merged = {}
        for g in gathered:
            assert not (set(g) & set(merged))
            merged.update(g)
        oracle = Oracle()
        want = {}
        for f in range(N_FRAMES):
            img = HostImage(W, H)
            st, _ = oracle.ljpeg(descs[f], datas[f], img)
            assert st == 0
            want[f] = G.image_hash(img.pixels())
        ok = merged == want and int(n_total) == N_FRAMES and t_max == float(grp.world)
        with open(sys.argv[1], "w") as fo:
            json.dump({"ok": bool(ok), "frames": len(merged), "world": grp.world,
                       "backend": grp.dist.get_backend()}, fo)
    grp.close()


if __name__ == "__main__":
    main()
