"""More seeds of the differential fuzz tests than the suite runs (GPU box):
   python scripts/fuzz_more.py two_tables 24 224   |   single 48 248   |   big 0 40   |   big2 0 40
   |   big3 0 40 (3 components)   |   bigdri 0 40 (restart intervals, 1-4 components)"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util
from oracle_lib import Oracle
which, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
gpu, oracle = gpu_util.ctx(), Oracle()
if which == "two_tables":
    import test_gpu_two_tables as T
    fn = T.test_fuzz_two_tables
elif which == "single":
    import test_gpu_fast_fuzz as T
    fn = T.test_fuzz_single_pass_kernel
else:
    fn = None
bad = []
for seed in (range(lo, hi) if fn else []):
    try:
        fn(gpu, oracle, seed)
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", str(e)[:300].replace("\n", " "))
if fn:
    print(which, "seeds", lo, "..", hi - 1, "failed:", bad)


def big(seed, two):
    """the same generators at sizes with hundreds of workgroups per stream"""
    import numpy as np
    import cases as C
    from rawspeed_amd import abi, synth
    from oracle_lib import HostImage
    from test_gpu_fast_fuzz import banded_image
    rng = np.random.default_rng([4041, seed])
    n = int(rng.choice([2, 2, 4]))
    prec = int(rng.choice([12, 14, 14, 16]))
    n_cat = 17 if prec == 16 else prec + 1
    ta = C.random_huffman_table(rng, n_cat, skew=float(rng.uniform(0.4, 2.5)))
    tb = C.random_huffman_table(rng, n_cat, skew=float(rng.uniform(0.4, 2.5))) if two else ta
    k = int(rng.integers(1, 4))
    H = int(rng.integers(900, 2400))
    tiles, x = [], 0
    for _ in range(k):
        tw = n * int(rng.integers(300, 4200 // n))
        tiles.append((x, tw)); x += tw
    W = x + int(rng.integers(0, 9))
    img, want = HostImage(W, H), HostImage(W, H)
    descs, datas = [], []
    for tx, tw in tiles:
        th = H - int(rng.integers(0, 3))
        px = banded_image(rng, th, tw, prec)
        fw = (tw + n - 1) // n + int(rng.integers(0, 3))
        rows = C.ljpeg_stream_rows(px, n, 1, fw, th, rng, prec)
        init = [1 << (prec - 1)] * n
        idx = ([0, 1] if rng.integers(0, 2) else [1, 0]) * (n // 2) if two else [0] * n
        tabs = (ta, tb) if two else (ta,)
        scan, _ = synth.ljpeg_encode_scan(rows, n, init, [tabs[i] for i in idx], 0, False)
        d = abi.LJpegDesc()
        d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, 0, tw, th
        d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = n, 1, fw, th
        d.n_comp, d.rows_per_restart_interval = n, th
        abi.fill_recipe(d, synth.huff_tables(*tabs), idx, init)
        descs.append(d)
        datas.append(np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(64, np.uint8)]))
    so = [oracle.ljpeg(d, data, want) for d, data in zip(descs, datas)]
    rc, st, cons = gpu.dng_decompress_ljpeg(descs, datas, img.view())
    assert [s[0] for s in so] == list(st), (st, so)
    assert all(c == s[1] for c, s in zip(cons, so) if s[0] == 0), (cons, so)
    if all(s[0] == 0 for s in so):
        assert np.array_equal(img.u16(), want.u16())


def big_r05(seed, dri):
    """3 components / restart intervals (tests/test_gpu_fuzz_r05.py) at the same sizes"""
    import numpy as np
    import test_gpu_fuzz_r05 as T5
    rng = np.random.default_rng([4042, seed, 1 if dri else 0])
    if not dri:
        return T5._run(gpu, oracle, rng, 3, 3, lambda th: 0, big=True)
    n = int(rng.choice([1, 2, 2, 3, 4]))
    return T5._run(gpu, oracle, rng, n, 3 if n == 3 else 1,
                   lambda th: int(rng.integers(1, max(2, th // 2))) if rng.integers(0, 5) else 0, big=True)


if which.startswith("big"):
    bad = []
    for seed in range(lo, hi):
        try:
            if which in ("big3", "bigdri"):
                big_r05(seed, which == "bigdri")
            else:
                big(seed, which == "big2")
        except Exception as e:  # noqa: BLE001
            bad.append(seed)
            print("seed", seed, "FAILED:", str(e)[:300].replace("\n", " "))
    print(which, "seeds", lo, "..", hi - 1, "failed:", bad)
