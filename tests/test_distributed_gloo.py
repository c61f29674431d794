"""N>1 path on CPU: world_size 2, gloo backend (the GPU run uses the same code
with backend nccl = RCCL)."""
import json
import os
import subprocess
import sys

from rawspeed_amd import dist

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_covers_everything_once():
    for n in (1, 7, 8, 255, 256):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = dist.shard_range(n, world, r)
                assert 0 <= lo <= hi <= n
                got.extend(range(lo, hi))
            assert got == list(range(n))
            sizes = [dist.shard_range(n, world, r) for r in range(world)]
            assert max(h - l for l, h in sizes) - min(h - l for l, h in sizes) <= 1
    assert dist.shard_range(256, 8, 3) == (96, 128)   # BASELINE config 5


import pytest


@pytest.mark.parametrize("world,port", [(2, 29533), (3, 29535)])
def test_gloo_run(tmp_path, world, port):
    """sharded decode, broadcast, and the cfg-5 distribution code (dist.distribute_units:
    scatter of per-rank shards that differ in content and size) at world 2 and 3"""
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(HERE, "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    res = json.loads(out.read_text())
    assert res == {"ok": True, "frames": 7, "t_max": 0.5 + world - 1, "world": world}


def test_scatter_of_identical_buffer_would_be_caught():
    """the round-3 defect: rank 0 sent ITS shard to everybody.  With units that differ in
    size the per-rank shard sizes differ, so a receiver sized for its own shard cannot even
    take rank 0's -- the size table distribute_units works from shows it."""
    sizes = [1000 + 137 * ((g * 5) % 7) for g in range(11)]
    per_rank = [sum(sizes[lo:hi]) for lo, hi in (dist.shard_range(11, 3, r) for r in range(3))]
    assert len(set(per_rank)) == 3
