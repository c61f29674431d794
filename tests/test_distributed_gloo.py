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


def test_two_rank_gloo_run(tmp_path):
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(HERE, "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    res = json.loads(out.read_text())
    assert res == {"ok": True, "frames": 7, "t_max": 1.5, "world": 2}
