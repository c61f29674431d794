"""Seeded differential fuzzing of the entropy pipeline against the oracle:
random canonical Huffman tables (every code-length shape JPEG allows), random
geometry, random noise levels (from nearly constant, where the bit stream barely
self-synchronises, to white noise full of FF00 stuffing), restart intervals, CR2
slicing, and random damage."""
import numpy as np
import pytest

from rawspeed_amd import abi

import cases as C
from oracle_lib import HostImage

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


def _tables(rng, n, prec):
    n_cat = 17 if prec == 16 else prec + 1
    k = int(rng.integers(1, n + 1))
    tables = tuple(C.random_huffman_table(rng, n_cat) for _ in range(k))
    return tables, [int(rng.integers(0, k)) for _ in range(n)]


@pytest.mark.parametrize("seed", range(160))
def test_fuzz_ljpeg(gpu, oracle, seed):
    rng = np.random.default_rng([2024, seed])
    mw, mh = [(1, 1), (2, 1), (3, 1), (4, 1), (2, 2), (3, 1)][seed % 6]
    cpp = 3 if seed % 6 == 5 else 1
    prec = int(rng.choice([8, 12, 14, 16]))
    th = mh * int(rng.integers(1, 200))
    tw = int(rng.integers(mw, 900 // cpp))
    fw = (cpp * tw + mw - 1) // mw + int(rng.integers(0, 3))      # frame wider than the tile
    fh = th // mh + int(rng.integers(0, 2))
    tx, ty = int(rng.integers(0, 40)), int(rng.integers(0, 40))
    W, H = tx + tw + int(rng.integers(0, 8)), ty + th + int(rng.integers(0, 8))
    tables, index = _tables(rng, mw * mh, prec)
    full = bool(rng.random() < 0.25)
    sigma = float(rng.choice([0.0, 0.5, 3.0, 30.0, 300.0]))
    ri = 0 if rng.random() < 0.6 else int(rng.integers(1, fh + 1))
    fix16 = bool(prec == 16 and rng.random() < 0.5)
    d, data, tile_px, _ = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=cpp, tile=(tx, ty, tw, th), mcu=(mw, mh),
        frame=(fw, fh), tables=tables, table_index=index, rows_per_ri=ri, fix16=fix16,
        prec=prec, full_range=full, sigma=sigma)
    img, want = HostImage(W, H, cpp), HostImage(W, H, cpp)
    so = oracle.ljpeg(d, data, want)
    sg = gpu.ljpeg_decode(d, data, img.view())
    assert sg == so, (sg, so)
    if so[0] == 0:
        assert np.array_equal(img.u16(), want.u16())
        assert np.array_equal(img.pixels()[ty:ty + th, cpp * tx:cpp * (tx + tw)], tile_px)


@pytest.mark.parametrize("seed", range(64))
def test_fuzz_cr2(gpu, oracle, seed):
    rng = np.random.default_rng([2025, seed])
    n = int(rng.choice([2, 4]))
    prec = int(rng.choice([12, 14]))
    h = int(rng.integers(2, 300))
    num = int(rng.integers(1, 5))
    sw = n * int(rng.integers(4, 120)) if num > 1 else 0
    last = n * int(rng.integers(4, 160))
    w = sw * (num - 1) + last
    tables, index = _tables(rng, n, prec)
    sigma = float(rng.choice([0.0, 1.0, 30.0, 300.0]))
    d, data, src, _ = C.make_cr2_case(rng, w, h, n, (num, sw, last), tables=tables,
                                      table_index=index, prec=prec, sigma=sigma,
                                      full_range=bool(rng.random() < 0.2))
    img, want = HostImage(w, h), HostImage(w, h)
    so = oracle.cr2(d, data, want)
    sg = gpu.cr2_decode(d, data, img.view())
    assert sg == so, (sg, so)
    if so[0] == 0:
        assert np.array_equal(img.u16(), want.u16())
        assert np.array_equal(img.pixels(), src)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_damaged(gpu, oracle, seed):
    """Random damage on random-table streams: same verdict as the reference
    algorithm, same pixels whenever it succeeds."""
    rng = np.random.default_rng([2026, seed])
    prec = int(rng.choice([12, 14, 16]))
    tables, index = _tables(rng, 2, prec)
    W, H = 640, 120
    d, data, _, scan_len = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1), tables=tables,
        table_index=index, prec=prec, sigma=float(rng.choice([2.0, 40.0])))
    for trial in range(12):
        bad = data.copy()
        kind = trial % 3
        if kind == 0:
            bad = bad[:int(rng.integers(8, scan_len))]
        elif kind == 1:
            idx = rng.integers(0, scan_len, size=int(rng.integers(1, 4)))
            bad[idx] = rng.integers(0, 256, size=idx.size)
        else:
            a = int(rng.integers(0, scan_len - 8))
            bad[a:a + int(rng.integers(1, 64))] = 0xFF
        img, want = HostImage(W, H), HostImage(W, H)
        so = oracle.ljpeg(d, bad, want)
        sg = gpu.ljpeg_decode(d, bad, img.view())
        if so[0] == 0:
            assert sg == so, (trial, sg, so)
            assert np.array_equal(img.u16(), want.u16()), trial
        else:
            assert sg[0] != 0, (trial, sg, so)
