"""The host-pointer path with rectangles that do NOT lie on the 16-byte grid of the caller's image --
the shape of round 5's one silent-wrong-pixels event (scripts/fuzz_more.py big3: 3-sample pixels,
tiles of 2-9 MB and unequal heights side by side, status OK, sixteen bytes of one row still holding
the caller's fill; profiles/r05/fuzz_big_and_ragged_download.txt).

What the reference promises and these tests hold the library to: the rectangle a decompressor owns is
fully written and nothing outside it is touched (LJpegDecompressor.cpp:264-268 decodes into
`[offX, offX + w) x [offY, offY + h)`, AbstractDngDecompressor.cpp:112-131 hands every tile its own
offsets; UncompressedDecompressor.cpp:188-200 for the packed path).  Every image is compared WHOLE
with the oracle's -- padding bytes included (both start from the same 0xA5 fill).

Cases: 1, 2, 3 and 4 interleaved components over cpp 1 / 2 / 3 images; several tiles of unequal
heights; ONE tile in the middle of an image decoded by itself (one off-grid rectangle), below and
above 256 KB; odd widths; the same through rsx_ljpeg_decode, rsx_unpack_u16 (bps 16 copy path with a
crop, packed with odd widths) -- six host threads at once, as rawspeed's OpenMP tile loops call."""
import os
import threading

import numpy as np
import pytest

from rawspeed_amd import abi, synth

import cases as C
from oracle_lib import HostImage
from test_gpu_fast_fuzz import banded_image

pytestmark = pytest.mark.gpu

BASE = int(os.environ.get("RSX_FUZZ_BASE", "0"))
N_THREADS = 6


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


def _tile(rng, tx, ty, tw_px, th, n, cpp, prec, table):
    samples = tw_px * cpp
    px = banded_image(rng, th, samples, prec)
    fw = (samples + n - 1) // n + int(rng.integers(0, 3))
    rows = C.ljpeg_stream_rows(px, n, 1, fw, th, rng, prec)
    init = [1 << (prec - 1)] * n
    scan, _ = synth.ljpeg_encode_scan(rows, n, init, [table] * n, 0, False)
    d = abi.LJpegDesc()
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, ty, tw_px, th
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = n, 1, fw, th
    d.n_comp, d.rows_per_restart_interval = n, th
    abi.fill_recipe(d, synth.huff_tables(table), [0] * n, init)
    return d, np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(64, np.uint8)])


def make_case(seed, kind):
    """kind: 'ragged' (1-3 tiles of unequal heights side by side), 'middle' (one tile that starts
    off the left edge, decoded by itself), 'odd' (one tile from the left edge, odd width);
    sizes from a few KB to ~9 MB a rectangle."""
    rng = np.random.default_rng([6061, BASE, seed, {"ragged": 0, "middle": 1, "odd": 2}[kind]])
    n, cpp = [(1, 1), (2, 1), (4, 1), (2, 2), (3, 3), (3, 3)][int(rng.integers(0, 6))]
    prec = int(rng.choice([12, 14, 14, 16]))
    table = C.random_huffman_table(rng, 17 if prec == 16 else prec + 1, skew=float(rng.uniform(0.4, 2.5)))
    unit = n // cpp if n % cpp == 0 and n >= cpp else 1   # pixels per MCU
    size = int(rng.integers(0, 3))                        # 0: tens of KB, 1: hundreds, 2: MBs
    H = [int(rng.integers(20, 90)), int(rng.integers(120, 420)), int(rng.integers(700, 1800))][size]
    wmax = [400, 1400, 3600][size] // (cpp * unit)
    tiles, x = [], 0
    if kind == "ragged":
        for _ in range(int(rng.integers(2, 4))):
            tw = unit * int(rng.integers(max(8, wmax // 4), max(9, wmax)))
            tiles.append((x, tw, H - int(rng.integers(0, 3)), True))
            x += tw
        if len({t[2] for t in tiles}) == 1:               # (unequal heights is the point)
            tiles[-1] = tiles[-1][:2] + (H - 1 - (tiles[0][2] == H - 1), True)
    elif kind == "middle":
        lead = unit * int(rng.integers(1, 40))            # pixels in front of the tile: never decoded
        tw = unit * int(rng.integers(max(8, wmax // 4), max(9, wmax)))
        tiles = [(lead, tw, H - int(rng.integers(0, 2)), True)]
        x = lead + tw
    else:
        tw = unit * int(rng.integers(max(8, wmax // 4), max(9, wmax))) | (1 if unit == 1 else 0)
        tiles = [(0, tw, H, True)]
        x = tw
    W = x + int(rng.integers(0, 5))
    descs, datas = [], []
    for tx, tw, th, _ in tiles:
        d, data = _tile(rng, tx, 0, tw, th, n, cpp, prec, table)
        descs.append(d)
        datas.append(data)
    return dict(W=W, H=H, cpp=cpp, descs=descs, datas=datas, kind=kind, seed=seed)


def oracle_image(oracle, c):
    want = HostImage(c["W"], c["H"], c["cpp"], is_cfa=c["cpp"] == 1)
    so = [oracle.ljpeg(d, data, want) for d, data in zip(c["descs"], c["datas"])]
    assert all(s[0] == 0 for s in so), (c["kind"], c["seed"], so)
    return want, so


def run_threads(work, n_threads=N_THREADS):
    """work: list of callables; run on n_threads host threads, re-raise the first failure"""
    errs, it, lock = [], iter(work), threading.Lock()

    def worker():
        while True:
            with lock:
                fn = next(it, None)
            if fn is None or errs:
                return
            try:
                fn()
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
    ts = [threading.Thread(target=worker) for _ in range(n_threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]


def describe(c, img, want):
    got, ref = img.u16(), want.u16()
    bad = np.argwhere(got != ref)
    r, s = bad[0]
    # (undelivered bytes show the image's fill, 0xA5; wrong pixels anything else)
    fill = int(np.count_nonzero(got[got != ref] == 0xA5A5))
    rows = np.unique(bad[:, 0])
    return ("%s seed %d: image %dx%d cpp %d pitch %d, %d wrong samples (%d of them the image's fill 0xA5A5) in %d "
            "rows %s, first at row %d sample %d (got %d, want %d), last at row %d sample %d, tiles %s" % (
                c["kind"], c["seed"], c["W"], c["H"], c["cpp"], img.pitch, len(bad), fill, len(rows),
                rows[:8].tolist(), r, s, got[r, s], ref[r, s], bad[-1][0], bad[-1][1],
                [(d.tile_x, d.tile_w, d.tile_h) for d in c["descs"]]))


@pytest.mark.parametrize("kind,lo,hi", [("ragged", 0, 90), ("middle", 0, 70), ("odd", 0, 50)])
def test_ragged_rectangles_come_back_whole_from_six_threads(gpu, oracle, kind, lo, hi):
    """210 images over the three shapes; every image decoded on one of six host threads through
    rsx_dng_decompress_ljpeg while the others decode theirs; whole buffers against the oracle."""
    cases = [make_case(s, kind) for s in range(lo, hi)]
    wants = [oracle_image(oracle, c) for c in cases]
    results = [None] * len(cases)

    def job(i):
        def run():
            c = cases[i]
            img = HostImage(c["W"], c["H"], c["cpp"], is_cfa=c["cpp"] == 1)
            rc, st, cons = gpu.dng_decompress_ljpeg(c["descs"], c["datas"], img.view())
            results[i] = (img, rc, st, cons)
        return run
    run_threads([job(i) for i in range(len(cases))])
    for c, (want, so), (img, rc, st, cons) in zip(cases, wants, results):
        assert rc == 0 and list(st) == [0] * len(st), (c["kind"], c["seed"], rc, st)
        assert list(cons) == [s[1] for s in so], (c["kind"], c["seed"], cons, so)
        assert np.array_equal(img.buf, want.buf), describe(c, img, want)


@pytest.mark.parametrize("seed", range(24))
def test_single_tile_calls_off_the_grid(gpu, oracle, seed):
    """rsx_ljpeg_decode (LJpegDecoder::decode's call) with one rectangle in the middle of the image"""
    c = make_case(1000 + seed, "middle")
    want, so = oracle_image(oracle, c)
    img = HostImage(c["W"], c["H"], c["cpp"], is_cfa=c["cpp"] == 1)
    st, cons = gpu.ljpeg_decode(c["descs"][0], c["datas"][0], img.view())
    assert (st, cons) == so[0]
    assert np.array_equal(img.buf, want.buf), describe(c, img, want)


@pytest.mark.parametrize("seed", range(16))
def test_unpack_off_the_grid(gpu, oracle, seed):
    """UncompressedDecompressor through the host-pointer call: widths whose rows end off the
    16-byte grid (packed 12 / 14 bit), and the 16-bit copy path with a crop (the only unpack
    shape whose rectangle STARTS off the grid); one frame large enough for the banded path."""
    rng = np.random.default_rng([6062, BASE, seed])
    big = seed % 4 == 3
    h = int(rng.integers(1500, 2600)) if big else int(rng.integers(8, 300))
    if seed % 2 == 0:
        bps, order = int(rng.choice([12, 14])), int(rng.choice([abi.ORDER_MSB, abi.ORDER_LSB]))
        w = (int(rng.integers(3000, 4400)) if big else int(rng.integers(40, 1200))) // 8 * 8 + 4
        while (w * bps) % 8:
            w += 4
        pitch = w * bps // 8 + int(rng.integers(0, 3)) * 4
        d = abi.UnpackDesc(0, 0, w, h, pitch, bps, order)
        W = w + int(rng.integers(0, 4))
    else:
        bps, order = 16, abi.ORDER_LSB
        w = (int(rng.integers(3000, 4400)) if big else int(rng.integers(40, 1200))) | 1
        cx = int(rng.integers(1, 9))
        pitch = w * 2
        d = abi.UnpackDesc(cx, 0, w, h, pitch, bps, order)
        W = cx + w + int(rng.integers(0, 4))
    data = rng.integers(0, 256, size=h * pitch, dtype=np.uint8)
    got, want = HostImage(W, h), HostImage(W, h)
    so = oracle.unpack(d, data, want)
    sg = gpu.unpack_u16(d, data, got.view())
    assert sg == so == 0, (sg, so, bps, w, h)
    assert np.array_equal(got.buf, want.buf), (bps, order, w, h, pitch, W)


def small_cases(oracle, n_cases, key=0):
    """single small tiles (a few KB of scan: a plan's kernels start microseconds after it is made)"""
    rng = np.random.default_rng([6062, BASE, key])
    cases = []
    for s in range(n_cases):
        n, cpp = [(1, 1), (2, 1), (2, 2), (3, 3)][int(rng.integers(0, 4))]
        prec = int(rng.choice([12, 14]))
        table = C.random_huffman_table(rng, prec + 1, skew=float(rng.uniform(0.6, 2.0)))
        unit = n // cpp if n % cpp == 0 and n >= cpp else 1
        tw, th = unit * int(rng.integers(24, 200)), int(rng.integers(16, 60))
        lead = unit * int(rng.integers(0, 9))
        d, data = _tile(rng, lead, 0, tw, th, n, cpp, prec, table)
        c = dict(W=lead + tw + int(rng.integers(0, 4)), H=th, cpp=cpp, descs=[d], datas=[data], kind="small", seed=s)
        cases.append((c,) + oracle_image(oracle, c))
    return cases


def busy_null_stream(stop, burst=4, cycles=24000):
    """keeps the null stream busy with short kernels, never more than `burst` of them queued (torch's
    default stream is the null stream)"""
    import torch
    ev = torch.cuda.Event()
    while not stop.is_set():
        for _ in range(burst):
            torch.cuda._sleep(cycles)
        ev.record()
        ev.synchronize()


def run_against_busy_null_stream(gpu, cases, rounds, burst=4, cycles=24000):
    import torch
    stop = threading.Event()
    bg = threading.Thread(target=busy_null_stream, args=(stop, burst, cycles), daemon=True)
    bg.start()
    wrong, calls = [], 0
    try:
        for rnd in range(rounds):
            for c, want, so in cases:
                img = HostImage(c["W"], c["H"], c["cpp"], is_cfa=c["cpp"] == 1)
                rc, st, cons = gpu.dng_decompress_ljpeg(c["descs"], c["datas"], img.view())
                calls += 1
                if not (rc == 0 and list(st) == [0] and list(cons) == [so[0][1]] and np.array_equal(img.buf, want.buf)):
                    wrong.append((rnd, c["seed"], rc, list(st), list(cons), so[0][1]))
    finally:
        stop.set()
        bg.join()
        torch.cuda.synchronize()
    return calls, wrong


def test_plan_creation_against_a_busy_null_stream(gpu, oracle):
    """hipMemset of device memory is asynchronous with respect to the host on this runtime, and a
    hipStreamNonBlocking stream is not ordered behind the null stream (scripts/repro/memset_async.hip).
    Until round 6 a plan's creation zeroed six device arrays with it and returned; with other host
    threads' work queued on the null stream the memsets ran under the plan's first run, and about once
    in 17 000 calls from six threads a whole small tile came back wrong with status OK
    (profiles/r06/host_path_defect/memset_async_and_the_wrong_tiles.txt).  Here a second thread keeps
    the null stream busy with short kernels while new plans of small tiles are made and run one after
    the other (scripts/exp_null_stream_stress.py has the rates of the library as it was)."""
    # (bursts of eight 10-us kernels, 1200 calls: the library as it was gets 2-3 % of them wrong)
    calls, wrong = run_against_busy_null_stream(gpu, small_cases(oracle, 100), 12, burst=8, cycles=24000)
    assert not wrong, (calls, len(wrong), wrong[:4])
