"""One seed of tests/test_gpu_two_tables.py::test_fuzz_two_tables, verbosely."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases as C
from rawspeed_amd import abi, synth, capi
from oracle_lib import HostImage, Oracle
from test_gpu_fast_fuzz import banded_image
seed = int(os.environ.get("SEED", "2"))
gpu = capi.Context(0); oracle = Oracle()
rng = np.random.default_rng([3031, seed])
n = int(rng.choice([2, 2, 4]))
prec = int(rng.choice([12, 14, 14, 16]))
n_cat = 17 if prec == 16 else prec + 1
ta = C.random_huffman_table(rng, n_cat, skew=float(rng.uniform(0.4, 2.5)))
tb = C.random_huffman_table(rng, n_cat, skew=float(rng.uniform(0.4, 2.5)))
k = int(rng.integers(1, 4))
tiles, x = [], 0
H = int(rng.integers(120, 500))
for _ in range(k):
    tw = n * int(rng.integers(40, 1400 // n))
    tiles.append((x, tw)); x += tw
W = x + int(rng.integers(0, 9))
img, want = HostImage(W, H), HostImage(W, H)
descs, datas, pxs = [], [], []
for tx, tw in tiles:
    th = H - int(rng.integers(0, 3))
    px = banded_image(rng, th, tw, prec)
    fw = (tw + n - 1) // n + int(rng.integers(0, 3))
    rows = C.ljpeg_stream_rows(px, n, 1, fw, th, rng, prec)
    init = [1 << (prec - 1)] * n
    order = [0, 1] if rng.integers(0, 2) else [1, 0]
    idx = order * (n // 2)
    scan, _ = synth.ljpeg_encode_scan(rows, n, init, [(ta, tb)[i] for i in idx], 0, False)
    d = abi.LJpegDesc()
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, 0, tw, th
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = n, 1, fw, th
    d.n_comp, d.rows_per_restart_interval = n, th
    abi.fill_recipe(d, synth.huff_tables(ta, tb), idx, init)
    tail = int(rng.integers(0, 3))
    extra = {0: np.zeros(16, np.uint8), 1: np.zeros(int(rng.integers(16, 40000)), np.uint8),
             2: rng.integers(0, 256, int(rng.integers(16, 40000)), dtype=np.uint8)}[tail]
    descs.append(d); datas.append(np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), extra])); pxs.append(px)
    print("tile", tx, tw, th, "n", n, "prec", prec, "scan", len(scan), "tail", tail, len(extra), "idx", idx)
so = [oracle.ljpeg(d, data, want) for d, data in zip(descs, datas)]
rc, st, cons = gpu.dng_decompress_ljpeg(descs, datas, img.view())
print("oracle", so); print("gpu", rc, st, cons)
print("pixels equal", np.array_equal(img.u16(), want.u16()))
