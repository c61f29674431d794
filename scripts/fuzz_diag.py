"""Diagnose a failing seed of scripts/fuzz_more.py big3 / bigdri: statuses, consumed bytes, where the
first wrong sample lies (GPU box).   python scripts/fuzz_diag.py big3 10 [12 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpu_util
from oracle_lib import Oracle
import test_gpu_fuzz_r05 as T5

which, seeds = sys.argv[1], [int(x) for x in sys.argv[2:]]
gpu, oracle = gpu_util.ctx(), Oracle()
POISON = int(os.environ.get("POISON_GB", "0"))


def poison():
    """Fill free device memory with a pattern and hand it back: what the library allocates next is
    not the zero pages of a fresh process (a kernel that reads what nobody wrote shows)."""
    if not POISON:
        return
    import torch
    ts = [torch.full((1 << 30,), 0xCD, dtype=torch.uint8, device="cuda") for _ in range(POISON)]
    torch.cuda.synchronize()
    del ts
    torch.cuda.empty_cache()


class Spy:
    """the context, recording what the test's one device call saw and returned"""
    def __init__(self, g):
        self.g = g
    def dng_decompress_ljpeg(self, descs, datas, view):
        self.descs, self.datas, self.view = descs, datas, view
        self.ret = self.g.dng_decompress_ljpeg(descs, datas, view)
        import ctypes
        n = view.pitch_bytes * view.dim_y
        self.out = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(view.data)).copy()
        self.geom = (view.dim_x, view.dim_y, view.cpp, view.pitch_bytes)
        return self.ret


for seed in seeds:
    dri = which == "bigdri"
    rng = np.random.default_rng([4042, seed, 1 if dri else 0])
    spy = Spy(gpu)
    poison()
    # (the oracle's image is built inside _run: rebuild it here through a second oracle pass)
    try:
        if not dri:
            T5._run(spy, oracle, rng, 3, 3, lambda th: 0, big=True)
        else:
            n = int(rng.choice([1, 2, 2, 3, 4]))
            T5._run(spy, oracle, rng, n, 3 if n == 3 else 1,
                    lambda th: int(rng.integers(1, max(2, th // 2))) if rng.integers(0, 5) else 0, big=True)
        print("seed", seed, "ok")
        continue
    except AssertionError as e:
        print("seed", seed, "FAILED", str(e)[:200])
    from oracle_lib import HostImage
    W, H, cpp, pitch = spy.geom
    want = HostImage(W, H, cpp, is_cfa=cpp == 1)
    so = [oracle.ljpeg(d, data, want) for d, data in zip(spy.descs, spy.datas)]
    rc, st, cons = spy.ret
    print("  image", W, "x", H, "cpp", cpp, "rc", rc, "statuses", list(st), "oracle", so, "consumed", list(cons))
    got = spy.out.view(np.uint16).reshape(H, pitch // 2)
    ref = want.u16()
    for d in spy.descs:
        x0, x1 = d.tile_x * cpp, (d.tile_x + d.tile_w) * cpp
        bad = np.argwhere(got[:d.tile_h, x0:x1] != ref[:d.tile_h, x0:x1])
        print("  tile x", d.tile_x, "w", d.tile_w, "h", d.tile_h, "frame_w", d.frame_w, "prec?", "wrong samples", len(bad),
              "first (row, sample)", bad[0].tolist() if len(bad) else None,
              "last", bad[-1].tolist() if len(bad) else None)
        if len(bad):
            r, c = bad[0]
            print("    got", got[r, x0 + c:x0 + c + 6].tolist(), "want", ref[r, x0 + c:x0 + c + 6].tolist(),
                  "rows with errors", len(np.unique(bad[:, 0])), "sample % 3 of errors", np.bincount(bad[:, 1] % 3, minlength=3).tolist())
