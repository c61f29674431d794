"""Differential fuzzing of what round 5 put on the single-pass kernel, against the oracle:
three interleaved components (MCU 3 x 1, cpp 3 images) and restart intervals laid out on the
device (any number of components, any interval height, several jobs per call, junk behind the
end-of-image marker), with random canonical tables and the banded images of
test_gpu_fast_fuzz (sensor noise, constant and clipped regions, ramps, short periods)."""
import os

import numpy as np
import pytest

from rawspeed_amd import abi, synth

import cases as C
from oracle_lib import HostImage
from test_gpu_fast_fuzz import banded_image

pytestmark = pytest.mark.gpu

# RSX_FUZZ_BASE=<k> moves every case to another seed (soak runs: scripts/rounds/r05/r05o.sh)
BASE = int(os.environ.get("RSX_FUZZ_BASE", "0"))


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


def _stream(rng, tx, tw_px, th, n, cpp, prec, table, rows_per_ri, tail):
    """a tile of tw_px pixels x th rows; n components per MCU row; cpp samples per pixel"""
    samples = tw_px * cpp
    px = banded_image(rng, th, samples, prec)
    fw = (samples + n - 1) // n + int(rng.integers(0, 3))   # frame wider than the tile (MCUs)
    rows = C.ljpeg_stream_rows(px, n, 1, fw, th, rng, prec)
    init = [1 << (prec - 1)] * n
    scan, _ = synth.ljpeg_encode_scan(rows, n, init, [table] * n, rows_per_ri, False)
    d = abi.LJpegDesc()
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, 0, tw_px, th
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = n, 1, fw, th
    d.n_comp = n
    d.rows_per_restart_interval = rows_per_ri if rows_per_ri else th
    abi.fill_recipe(d, synth.huff_tables(table), [0] * n, init)
    extra = {0: np.zeros(16, np.uint8), 1: np.zeros(int(rng.integers(16, 20000)), np.uint8),
             2: rng.integers(0, 256, int(rng.integers(16, 20000)), dtype=np.uint8),
             3: np.tile(np.array([0xFF, 0xE1, 0x24], np.uint8), int(rng.integers(6, 3000)))}[tail]
    return d, np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), extra]), px


def _run(gpu, oracle, rng, n, cpp, rows_per_ri_of, big=False):
    """big: hundreds of workgroups per stream (scripts/fuzz_more.py)"""
    prec = int(rng.choice([12, 14, 14, 16]))
    n_cat = 17 if prec == 16 else prec + 1
    table = C.random_huffman_table(rng, n_cat, skew=float(rng.uniform(0.4, 2.5)))
    k = int(rng.integers(1, 4))
    H = int(rng.integers(900, 2400)) if big else int(rng.integers(60, 420))
    tiles, x = [], 0
    unit = n // cpp if n % cpp == 0 and n >= cpp else 1   # pixels per MCU
    for _ in range(k):
        tw = unit * int(rng.integers(300 if big else 24,
                                     max(301 if big else 25, (4200 if big else 1500) // (cpp * unit))))
        tiles.append((x, tw))
        x += tw
    W = x + int(rng.integers(0, 5))
    img = HostImage(W, H, cpp, is_cfa=cpp == 1)
    want = HostImage(W, H, cpp, is_cfa=cpp == 1)
    descs, datas = [], []
    for tx, tw in tiles:
        th = H - int(rng.integers(0, 3))
        d, data, _ = _stream(rng, tx, tw, th, n, cpp, prec, table, rows_per_ri_of(th),
                             int(rng.integers(0, 4)))
        descs.append(d)
        datas.append(data)
    so = [oracle.ljpeg(d, data, want) for d, data in zip(descs, datas)]
    rc, st, cons = gpu.dng_decompress_ljpeg(descs, datas, img.view())
    for i in range(k):
        assert st[i] == so[i][0], (i, list(st), so)
        if so[i][0] == 0:
            assert cons[i] == so[i][1], (i, list(cons), so)
    if all(s[0] == 0 for s in so):
        assert np.array_equal(img.buf, want.buf)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_three_components(gpu, oracle, seed):
    rng = np.random.default_rng([2051, BASE, seed])
    _run(gpu, oracle, rng, 3, 3, lambda th: 0)


@pytest.mark.parametrize("seed", range(32))
def test_fuzz_restart_intervals(gpu, oracle, seed):
    rng = np.random.default_rng([2052, BASE, seed])
    n = int(rng.choice([1, 2, 2, 3, 4]))
    cpp = 3 if n == 3 else 1
    _run(gpu, oracle, rng, n, cpp,
         lambda th: int(rng.integers(1, max(2, th // 2))) if rng.integers(0, 5) else 0)
