"""A Huffman table PER COMPONENT on the single-pass kernel (round 6: LjStreamDev::fast == 3,
lj_fast_kernel<N, 2, .>, lj_unstuff_kernel<2, .>).

A DNG writer emits one DHT per component and names it in the scan header
(AbstractLJpegDecoder.cpp:181-228 parseSOS, :230-291 parseDHT; AbstractLJpegDecoder.h:112-125 binds
`huff[i]` per component; LJpegDecompressor.cpp:102-113 builds "one recipe per component",
:184-251 decodes component i of an MCU with `ht[i]`): a linear (3-sample) DNG therefore has THREE
tables, A B C.  Until round 5 such streams took the legacy route (stream-ordered int16 differences +
reconstruction passes), and so did every 4-component assignment other than A B A B.  Here: A B C,
A B B, A A B over three components; A B C D, A A A B, A A B B, A B B A over four; and a plan that
mixes them with one- and two-table streams.  Everything against the oracle, bit by bit; the route is
asserted from the kernels that ran."""
import numpy as np
import pytest
import torch

import cases as C
from oracle_lib import HostImage
from rawspeed_amd import abi, synth
from test_gpu_fast_fuzz import banded_image

pytestmark = pytest.mark.gpu

# RSX_FUZZ_BASE=<k> moves the fuzz cases to other seeds (soak runs)
import os
BASE = int(os.environ.get("RSX_FUZZ_BASE", "0"))


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


def _kernel_names(plan, inp, out):
    s = torch.cuda.current_stream().cuda_stream
    plan.set_timing(True)
    plan.run(inp.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    tab = plan.kernel_table()
    plan.set_timing(False)
    return [n for n, _ in tab[0]] if tab else []


def assert_single_pass(names):
    assert any("lj_fast_kernel" in x for x in names), names
    bad = [x for x in names if "sync" in x or "decode" in x or "legacy" in x or "rowedge" in x]
    assert not bad, names


def _tables(rng, k, prec):
    n_cat = 17 if prec == 16 else prec + 1
    return [C.random_huffman_table(rng, n_cat, skew=float(rng.uniform(0.5, 2.5))) for _ in range(k)]


def _stream(rng, px, n, prec, tables, index, tx=0, ty=0, pad=0):
    """px: (rows, samples) of one tile whose MCU is n x 1"""
    th, samples = px.shape
    fw = (samples + n - 1) // n + pad
    rows = C.ljpeg_stream_rows(px, n, 1, fw, th, rng, prec)
    init = [1 << (prec - 1)] * n
    scan, _ = synth.ljpeg_encode_scan(rows, n, init, [tables[i] for i in index], 0, False)
    d = abi.LJpegDesc()
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = n, 1, fw, th
    d.n_comp, d.rows_per_restart_interval = n, th
    abi.fill_recipe(d, synth.huff_tables(*tables), index, init)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(32, np.uint8)])
    return d, data


def _decode(gpu, oracle, items, W, H, cpp):
    """items: [(desc with tile_x/y/w/h set, data)] sharing one W x H x cpp image"""
    import bench_ljpeg as B
    want = HostImage(W, H, cpp, is_cfa=cpp == 1)
    so = [oracle.ljpeg(d, data, want) for d, data in items]
    assert all(s[0] == 0 for s in so), so
    op = B.out_pitch(W * cpp)
    jobs, off = [], 0
    for d, data in items:
        j = abi.LJpegJob()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = off, data.size, 0
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = op, W, H, cpp, int(cpp == 1)
        jobs.append(j)
        off += (data.size + 15) // 16 * 16
    buf = np.zeros(off + 64, np.uint8)
    for j, (d, data) in zip(jobs, items):
        buf[j.in_offset:j.in_offset + data.size] = data
    plan = gpu.ljpeg_plan(jobs)
    inp = torch.from_numpy(buf).cuda()
    out = torch.zeros(op * H, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        out.zero_()
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        assert rc == 0 and not any(st), (rc, st)
        assert list(cons) == [x[1] for x in so]
        got = out.cpu().numpy().view(np.uint16).reshape(H, op // 2)
        ref = want.u16()
        for d, _ in items:  # (the device image starts from zeros, the oracle's from its fill)
            x0, x1 = d.tile_x * cpp, (d.tile_x + d.tile_w) * cpp
            a, b = got[d.tile_y:d.tile_y + d.tile_h, x0:x1], ref[d.tile_y:d.tile_y + d.tile_h, x0:x1]
            assert np.array_equal(a, b), np.argwhere(a != b)[:4]
    return plan, inp, out


def _sensor_like(rng, h, w, lo=1500, hi=12000):
    """noise on slow ramps, well inside the 14-bit range: no clipped (constant) stretches -- whose
    zero codes of two or three bits are more symbols a slot than the kernel's lanes keep, the
    multi-kernel pipeline's business whatever the tables"""
    base = rng.uniform(lo, hi)
    img = base + 0.4 * (np.arange(w)[None, :] % 3000) + 0.9 * np.arange(h)[:, None] + rng.normal(0, 25.0, (h, w))
    return np.clip(img, 64, 16000).astype(np.uint16)


PATTERNS = [(3, 3, [0, 1, 2]), (3, 3, [0, 1, 1]), (3, 3, [0, 0, 1]), (3, 3, [2, 0, 1]),
            (4, 1, [0, 1, 2, 3]), (4, 1, [0, 0, 0, 1]), (4, 1, [0, 0, 1, 1]), (4, 1, [0, 1, 1, 0]),
            (4, 1, [0, 1, 2, 0])]


@pytest.mark.parametrize("n,cpp,index", PATTERNS, ids=["".join("ABCD"[i] for i in p[2]) + "_%d" % p[0] for p in PATTERNS])
def test_table_per_component_takes_the_single_pass_kernel(gpu, oracle, n, cpp, index):
    rng = np.random.default_rng([808, n] + index)
    prec = 14
    tw, th = 1536 if cpp == 1 else 768, 700
    tables = _tables(rng, max(index) + 1, prec)
    px = _sensor_like(rng, th, tw * cpp)
    d, data = _stream(rng, px, n, prec, tables, index)
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = 0, 0, tw, th
    assert d.n_tables == max(index) + 1
    plan, inp, out = _decode(gpu, oracle, [(d, data)], tw, th, cpp)
    assert_single_pass(_kernel_names(plan, inp, out))


@pytest.mark.parametrize("seed", range(20))
def test_fuzz_tables_per_component(gpu, oracle, seed):
    """random tables (long codes included), 12 / 14 / 16 bit, banded images with constant and clipped
    regions (the table-per-phase parse leaves those to the multi-kernel pipeline: any route, the
    reference's pixels), frames wider than their tiles, one to three tiles a call"""
    rng = np.random.default_rng([809, BASE, seed])
    n, cpp = (3, 3) if seed % 2 == 0 else (4, 1)
    prec = int(rng.choice([12, 14, 14, 16]))
    nt = int(rng.integers(2, n + 1))
    index = [int(rng.integers(0, nt)) for _ in range(n)]
    index[int(rng.integers(0, n))] = nt - 1          # (every table used ...)
    if len(set(index)) == 1:
        index[0] = (index[0] + 1) % nt               # (... and more than one)
    index = [sorted(set(index)).index(i) for i in index]
    tables = _tables(rng, max(index) + 1, prec)
    unit = n // cpp if n % cpp == 0 and n >= cpp else 1
    H = int(rng.integers(80, 500))
    items, x = [], 0
    for _ in range(int(rng.integers(1, 4))):
        tw = unit * int(rng.integers(40, 1400 // (cpp * unit)))
        th = H - int(rng.integers(0, 3))
        px = banded_image(rng, th, tw * cpp, prec) if seed % 4 >= 2 else \
            C.smooth_image(rng, th, tw * cpp, prec, sigma=float(rng.choice([3.0, 30.0, 300.0])),
                           full_range=prec == 16)
        d, data = _stream(rng, px, n, prec, tables, index, pad=int(rng.integers(0, 3)))
        d.tile_x, d.tile_y, d.tile_w, d.tile_h = x, 0, tw, th
        items.append((d, data))
        x += tw
    _decode(gpu, oracle, items, x + int(rng.integers(0, 4)), H, cpp)


def test_mixed_plan_one_two_and_three_tables(gpu, oracle):
    """one plan: a one-table stream, a two-alternating-table stream (which becomes a table-per-phase
    stream in such a plan) and an A B C D stream, all 4-component tiles of one image"""
    rng = np.random.default_rng(810)
    prec, th = 14, 600
    tabs = _tables(rng, 4, prec)
    items, x = [], 0
    for index in ([0, 0, 0, 0], [0, 1, 0, 1], [0, 1, 2, 3]):
        tw = 4 * int(rng.integers(200, 400))
        px = _sensor_like(rng, th, tw)
        used = sorted(set(index))
        d, data = _stream(rng, px, 4, prec, [tabs[i] for i in used], [used.index(i) for i in index])
        d.tile_x, d.tile_y, d.tile_w, d.tile_h = x, 0, tw, th
        items.append((d, data))
        x += tw
    plan, inp, out = _decode(gpu, oracle, items, x, th, 1)
    assert_single_pass(_kernel_names(plan, inp, out))


def test_host_pointer_call_linear_dng_tiles(gpu, oracle):
    """rsx_dng_decompress_ljpeg: 2 x 2 tiles of a 3-sample image, a table per component each"""
    rng = np.random.default_rng(811)
    prec, tw, th = 14, 640, 360
    W, H = 2 * tw, 2 * th
    descs, datas = [], []
    for ty in (0, th):
        for tx in (0, tw):
            tables = _tables(rng, 3, prec)
            px = C.smooth_image(rng, th, tw * 3, prec, sigma=15.0)
            d, data = _stream(rng, px, 3, prec, tables, [0, 1, 2])
            d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, ty, tw, th
            descs.append(d)
            datas.append(data)
    img, want = HostImage(W, H, 3, is_cfa=False), HostImage(W, H, 3, is_cfa=False)
    so = [oracle.ljpeg(d, data, want) for d, data in zip(descs, datas)]
    rc, st, cons = gpu.dng_decompress_ljpeg(descs, datas, img.view())
    assert rc == 0 and list(st) == [0] * 4 and list(cons) == [s[1] for s in so]
    assert np.array_equal(img.buf, want.buf)
