"""One large entropy-coded stream through a host-pointer call, in chunks: the upload of the next quarter
and the download of the finished rows under the decode (rsx_api.hip, ljpeg_chunked_host; round 6).

The callers are LJpegDecoder::decode (LJpegDecoder.cpp:161-164) and Cr2LJpegDecoder::decode
(Cr2LJpegDecoder.cpp:150-153): one frame, one scan.  What must hold is what holds for the plain way --
the rectangle the decompressor owns fully written, nothing outside it touched
(LJpegDecompressor.cpp:264-268, Cr2DecompressorImpl.h:437-468), status and consumed bytes the
reference's -- for whole stream rows, odd widths, tiles inside a wider image,
and for a stream that leaves the single-pass kernel half-way (everything comes down again)."""
import numpy as np
import pytest

from rawspeed_amd import abi, capi, synth

import cases as C
from oracle_lib import HostImage

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    return capi.Context(0)      # (a context of its own: its counters are this module's)


def _sensor(rng, h, w):
    img = rng.uniform(2000, 11000) + 0.3 * (np.arange(w)[None, :] % 2500) + 0.5 * np.arange(h)[:, None] \
        + rng.normal(0, 30.0, (h, w))
    return np.clip(img, 64, 16000).astype(np.uint16)


def _ljpeg_case(rng, W, H, tx, tw, n, cpp=1, tables=(C.NIKON,), index=None):
    px = _sensor(rng, H, tw * cpp)
    fw = (tw * cpp + n - 1) // n
    rows = C.ljpeg_stream_rows(px, n, 1, fw, H, rng, 14)
    init = [1 << 13] * n
    index = index or [0] * n
    scan, _ = synth.ljpeg_encode_scan(rows, n, init, [tables[i] for i in index], 0, False)
    d = abi.LJpegDesc()
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, 0, tw, H
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = n, 1, fw, H
    d.n_comp, d.rows_per_restart_interval = n, H
    abi.fill_recipe(d, synth.huff_tables(*tables), index, init)
    return d, np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(64, np.uint8)])


@pytest.mark.parametrize("shape", ["full_2comp", "odd_width_1comp", "tile_in_a_wider_image_3comp", "two_tables"])
def test_ljpeg_frame_in_chunks(gpu, oracle, shape):
    rng = np.random.default_rng([7001, hash(shape) & 0xFFFF])
    if shape == "full_2comp":
        W, H, tx, tw, n, cpp, kw = 4096, 2200, 0, 4096, 2, 1, {}
    elif shape == "odd_width_1comp":
        W, H, tx, tw, n, cpp, kw = 4101, 2100, 0, 4099, 1, 1, {}
    elif shape == "tile_in_a_wider_image_3comp":
        W, H, tx, tw, n, cpp, kw = 1500, 2100, 37, 1400, 3, 3, {}
    else:
        W, H, tx, tw, n, cpp, kw = 4096, 2200, 0, 4096, 2, 1, dict(tables=(C.NIKON, C.ALT), index=[0, 1])
    d, data = _ljpeg_case(rng, W, H, tx, tw, n, cpp, **kw)
    assert data.size >= 8 << 20
    want = HostImage(W, H, cpp, is_cfa=cpp == 1)
    so = oracle.ljpeg(d, data, want)
    assert so[0] == 0
    before = gpu.chunked_calls()
    for call in range(3):           # (the first builds the lane's plan; the others find it)
        img = HostImage(W, H, cpp, is_cfa=cpp == 1)
        assert gpu.ljpeg_decode(d, data, img.view()) == so
        assert np.array_equal(img.buf, want.buf), (shape, call)
    assert gpu.chunked_calls() - before == 2, "calls 2 and 3 should have run in chunks"


def test_cr2_frame_takes_the_plain_way(gpu, oracle):
    """CR2 slices: what a prefix of the stream completes is rows of a vertical STRIP -- narrow 2-D
    copies, slower than the whole frame in one piece (measured: profiles/r06/ab/chunked_host_calls.txt)
    --, so such frames are not run in chunks; same pixels either way"""
    rng = np.random.default_rng(7002)
    W, H = 5184, 1900
    d, data, img_px, _ = C.make_cr2_case(rng, W, H, 2, (3, 1728, 1728))
    assert data.size >= 8 << 20
    want = HostImage(W, H)
    so = oracle.cr2(d, data, want)
    assert so[0] == 0
    before = gpu.chunked_calls()
    for call in range(3):
        img = HostImage(W, H)
        assert gpu.cr2_decode(d, data, img.view()) == so
        assert np.array_equal(img.buf, want.buf), call
    assert gpu.chunked_calls() == before


def test_stream_that_leaves_the_single_pass_kernel_half_way(gpu, oracle):
    """the lane's plan is the frame's, the DATA of the second call is damaged behind its middle: what
    the chunks fetched early is fetched again after the second pass; status and partial image are
    whatever the oracle says for that stream"""
    rng = np.random.default_rng(7003)
    W, H = 4096, 2200
    d, data = _ljpeg_case(rng, W, H, 0, W, 2)
    bad = data.copy()
    mid = bad.size * 5 // 8
    bad[mid:mid + 4096] = rng.integers(0, 255, 4096, dtype=np.uint8)   # (no FF: no marker, garbage codes)
    for blob in (data, bad, data):
        want = HostImage(W, H)
        so = oracle.ljpeg(d, blob, want)
        img = HostImage(W, H)
        sg = gpu.ljpeg_decode(d, blob, img.view())
        assert sg[0] == so[0]
        if so[0] == 0:
            assert sg == so and np.array_equal(img.buf, want.buf)
