"""The N>1 code path on real GPUs: torch.distributed backend "nccl" (RCCL) at whatever
world size the box offers (1 on the single-GPU test box, up to 8 on a node): process
group, barrier, all-reduce, broadcast and grouped send/recv of the packed batch,
sharded decode through the C-ABI, gathered results."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _run(world, tmp_path, port):
    out = tmp_path / ("result_%d.json" % world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_worker_gpu.py"), str(out)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    return json.loads(out.read_text())


def test_nccl_sharded_decode_at_visible_world_size(tmp_path):
    import torch
    n = torch.cuda.device_count()
    assert n >= 1
    res = _run(1, tmp_path, 29541)
    assert res == {"ok": True, "frames": 6, "world": 1, "backend": "nccl"}
    if n >= 2:
        world = min(n, 8)
        res = _run(world, tmp_path, 29543)
        assert res == {"ok": True, "frames": 6, "world": world, "backend": "nccl"}


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus N` without a launcher starts N ranks itself (N = the
    GPUs of the box, at most 2 here) and reports n_gpus = N."""
    import torch
    n = min(torch.cuda.device_count(), 2)
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", str(n),
           "--steps", "3", "--warmup", "1", "--frames", "2", "--no-extra", "--no-cfg5",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == n and res["bit_exact"] is True
    assert len(res["per_rank_mpix_per_s"]) == n
    assert res["rccl_ranks"] == (n if n > 1 else 0)


@pytest.mark.parametrize("mode", ["", "--scatter", "--broadcast"])
def test_bench_cfg5_distribution_over_rccl(mode):
    """The cfg-5 distribution path on RCCL when the box has at least two GPUs: rank 0
    hands the packed batch out (grouped send/recv of the shards, or one broadcast of the
    whole batch whose shards differ), every rank decodes its shard bit-exactly.  Without a
    flag -- what the driver's scaling run is -- the ONE JSON line carries all three records
    (own shard, scatter, broadcast: kernel-only and distribution-inclusive scaling apart)."""
    import torch
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("one GPU: nothing to distribute (the gloo test covers the logic)")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", str(n),
           "--steps", "2", "--warmup", "1", "--frames", "1", "--no-extra", "--no-cpu-baseline",
           "--cfg5-total-frames", str(2 * n), "--cfg5-distinct", "3"] + ([mode] if mode else [])
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c5 = res["ljpeg"]["cfg5_batch_8192x5464"]
    assert res["n_gpus"] == n and c5["bit_exact"] is True
    assert c5["frames_on_this_rank"] == 2 and res["rccl_ranks"] == n
    d = c5["input_distribution"]
    want = {"": {"own_shard", "scatter", "broadcast"}, "--scatter": {"own_shard", "scatter"},
            "--broadcast": {"own_shard", "broadcast"}}[mode]
    assert set(d) == want, d
    for m in want - {"own_shard"}:
        assert d[m]["bytes"] > 0 and d[m]["ms"] > 0 and d[m]["delivers_the_ranks_own_shard"] is True
        assert d[m]["decoded_bit_exact"] is True
        assert d[m]["mpix_per_s_incl_distribution"] < d["own_shard"]["mpix_per_s_incl_distribution"]
