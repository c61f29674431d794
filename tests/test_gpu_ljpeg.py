"""GPU parity of the lossless-JPEG pipeline (LJpegDecompressor, Cr2Decompressor,
DNG tiles) through the C-ABI vs the oracle, the reference's golden hashes, and
round trips at the BASELINE sizes."""
import json
import os

import numpy as np
import pytest
import torch

from rawspeed_amd import abi, synth

import cases as C
import golden_cases as G
from oracle_lib import HostImage, out_pitch

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "golden_hashes.json")) as f:
    GOLD = json.load(f)


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


@pytest.mark.parametrize("c", G.LJPEG_CASES, ids=lambda c: c["name"])
def test_ljpeg_golden(gpu, oracle, c):
    d, data, (w, h, cpp), tile_px = G.build_ljpeg(c)
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    st, consumed = gpu.ljpeg_decode(d, data, img.view())
    g = GOLD["ljpeg"][c["name"]]
    assert (st, consumed) == oracle.ljpeg(d, data, want) == (g["status"], g["consumed"])
    assert np.array_equal(img.u16(), want.u16())
    assert G.image_hash(img.pixels()) == g["hash"]


@pytest.mark.parametrize("c", G.CR2_CASES, ids=lambda c: c["name"])
def test_cr2_golden(gpu, oracle, c):
    d, data, (w, h, cpp), src = G.build_cr2(c)
    cfa = "sraw" not in c
    img, want = HostImage(w, h, cpp, is_cfa=cfa), HostImage(w, h, cpp, is_cfa=cfa)
    st, consumed = gpu.cr2_decode(d, data, img.view())
    g = GOLD["cr2"][c["name"]]
    assert (st, consumed) == oracle.cr2(d, data, want) == (g["status"], g["consumed"])
    assert np.array_equal(img.u16(), want.u16())
    assert G.image_hash(img.pixels()) == g["hash"]


SHAPES = [
    # (img_w, img_h, cpp, tile, mcu, frame, kwargs) -- several workgroups per stream
    (1536, 384, 1, (0, 0, 1536, 384), (2, 1), None, {}),
    (1536, 384, 1, (0, 0, 1536, 384), (1, 1), None, {}),
    (1536, 384, 1, (0, 0, 1536, 384), (4, 1), None, {}),
    (1536, 384, 1, (0, 0, 1536, 384), (2, 2), None, {}),
    (1533, 386, 1, (0, 0, 1533, 386), (3, 1), None, {}),
    (700, 300, 3, (10, 20, 601, 250), (3, 1), (640, 256), {}),
    (1200, 500, 1, (512, 256, 688, 244), (2, 1), (512, 256), {}),       # overhang
    (1536, 384, 1, (0, 0, 1536, 384), (2, 1), None,
     dict(tables=(C.NIKON, C.ALT), table_index=[0, 1])),
    (1024, 256, 1, (0, 0, 1024, 256), (2, 2), None,
     dict(tables=(C.NIKON, C.ALT, C.FULL17), table_index=[0, 1, 2, 1])),
    (1024, 256, 1, (0, 0, 1024, 256), (2, 1), None,
     dict(tables=(C.FULL17,), full_range=True, prec=16)),                # long codes, SSSS 16
    (1024, 256, 1, (0, 0, 1024, 256), (2, 1), None,
     dict(tables=(C.FULL17,), full_range=True, prec=16, fix16=True)),
    (1024, 256, 1, (0, 0, 1024, 256), (2, 1), None, dict(full_range=True)),  # many FF00
]


@pytest.mark.parametrize("k", range(len(SHAPES)))
def test_ljpeg_shapes_vs_oracle(gpu, oracle, k):
    w, h, cpp, tile, mcu, frame, kw = SHAPES[k]
    rng = np.random.default_rng([99, k])
    d, data, tile_px, scan_len = C.make_ljpeg_case(rng, img_w=w, img_h=h, cpp=cpp, tile=tile,
                                                   mcu=mcu, frame=frame, **kw)
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    so = oracle.ljpeg(d, data, want)
    sg = gpu.ljpeg_decode(d, data, img.view())
    assert sg == so and so[0] == 0
    assert np.array_equal(img.u16(), want.u16())
    tx, ty, tw, th = tile
    assert np.array_equal(img.pixels()[ty:ty + th, cpp * tx:cpp * (tx + tw)], tile_px)


@pytest.mark.parametrize("n,slices", [(2, (1, 0, 2016)), (2, (3, 672, 672)), (4, (2, 1504, 512)),
                                      (2, (4, 480, 576))])
def test_cr2_shapes_vs_oracle(gpu, oracle, n, slices):
    rng = np.random.default_rng([98, n, slices[0]])
    d, data, src, _ = C.make_cr2_case(rng, 2016, 400, n, slices)
    img, want = HostImage(2016, 400), HostImage(2016, 400)
    so = oracle.cr2(d, data, want)
    sg = gpu.cr2_decode(d, data, img.view())
    assert sg == so and so[0] == 0
    assert np.array_equal(img.u16(), want.u16())
    assert np.array_equal(img.pixels(), src)


def test_cr2_wrapped_slices(gpu, oracle):
    """frame.y != dim.y: slices twice the image height wrap into two columns
    (the Canon double-height quirk, Cr2LJpegDecoder.cpp:80-87)."""
    rng = np.random.default_rng(97)
    d, data, src, _ = C.make_cr2_case(rng, 640, 200, 2, (1, 0, 640))
    d.frame_w, d.frame_h = 160, 400
    d.num_slices, d.slice_width, d.last_slice_width = 2, 320, 320
    img, want = HostImage(640, 200), HostImage(640, 200)
    so = oracle.cr2(d, data, want)
    assert so[0] == 0
    assert gpu.cr2_decode(d, data, img.view()) == so
    assert np.array_equal(img.u16(), want.u16())


@pytest.mark.parametrize("ysf,slices,dim_y,kw", [
    (1, (1, 0, 1296), 300, {}),
    (1, (3, 432, 432), 300, {}),
    (2, (3, 320, 224), 260, {}),
    (2, (4, 216, 216), 240, dict(dim_x=432, frame_y=120)),              # wrapped slices
    (1, (3, 432, 432), 200, dict(tables=(C.NIKON, C.ALT), table_index=(0, 1, 1))),
    (2, (2, 400, 464), 200, dict(tables=(C.FULL17,), full_range=True, prec=16)),
])
def test_cr2_sraw_vs_oracle(gpu, oracle, ysf, slices, dim_y, kw):
    """Canon sRaw <3,2,1> / <3,2,2>: groups of 4 / 6 samples, luma predicted
    along the group, chroma from the previous group, row predictors from the
    first group of the previous frame row (Cr2DecompressorImpl.h:431-465)."""
    rng = np.random.default_rng([96, ysf, slices[0], dim_y])
    d, data, src, _ = C.make_cr2_sraw_case(rng, ysf, slices, dim_y, **kw)
    h, w = src.shape
    img, want = HostImage(w, h, is_cfa=False), HostImage(w, h, is_cfa=False)
    so = oracle.cr2(d, data, want)
    sg = gpu.cr2_decode(d, data, img.view())
    assert sg == so and so[0] == 0
    assert np.array_equal(img.u16(), want.u16())
    assert np.array_equal(img.pixels(), src)


def test_cr2_sraw_corrupt_streams(gpu, oracle):
    rng = np.random.default_rng(6)
    d, data, src, _ = C.make_cr2_sraw_case(rng, 2, (3, 160, 128), 120)
    h, w = src.shape
    n_ok = n_fail = 0
    for trial in range(16):
        bad = data.copy()
        if trial % 4 == 0:
            bad = bad[:rng.integers(64, len(bad) // 2)]
        elif trial % 4 == 1:
            bad[rng.integers(0, len(bad) - 40)] = 0xFF
        else:
            idx = rng.integers(0, len(bad) - 40, size=3)
            bad[idx] = rng.integers(0, 256, size=3)
        img, want = HostImage(w, h, is_cfa=False), HostImage(w, h, is_cfa=False)
        so = oracle.cr2(d, bad, want)
        sg = gpu.cr2_decode(d, bad, img.view())
        assert sg[0] == so[0], (trial, sg, so)
        if so[0] == 0:
            n_ok += 1
            assert sg == so
            assert np.array_equal(img.u16(), want.u16())
        else:
            n_fail += 1
    assert n_ok and n_fail


def test_ljpeg_corrupt_streams(gpu, oracle):
    """Status parity on damaged streams; bit-exact pixels whenever the reference
    algorithm succeeds."""
    rng = np.random.default_rng(5)
    d, data, tile_px, _ = C.make_ljpeg_case(rng, img_w=1024, img_h=200, cpp=1,
                                            tile=(0, 0, 1024, 200), mcu=(2, 1))
    n_fail = n_ok = 0
    for trial in range(24):
        bad = data.copy()
        if trial % 4 == 0:
            bad = bad[:rng.integers(64, len(bad) // 2)]
        elif trial % 4 == 1:
            bad[rng.integers(0, len(bad) - 40)] = 0xFF
        else:
            idx = rng.integers(0, len(bad) - 40, size=3)
            bad[idx] = rng.integers(0, 256, size=3)
        img, want = HostImage(1024, 200), HostImage(1024, 200)
        so = oracle.ljpeg(d, bad, want)
        sg = gpu.ljpeg_decode(d, bad, img.view())
        if so[0] == 0:
            n_ok += 1
            assert sg == so, (trial, sg, so)
            assert np.array_equal(img.u16(), want.u16()), trial
        else:
            n_fail += 1
            assert sg[0] != 0, (trial, sg, so)
    assert n_fail > 0 and n_ok > 0


def test_dng_ljpeg_tiles(gpu, oracle):
    """AbstractDngDecompressor::decompressThread<7>: 2x2 tiles of an odd-sized
    image (right/bottom tiles overhang: trailing pixel + discard + early stop),
    decoded by one batched call."""
    rng = np.random.default_rng(6)
    W, H, tw, th = 1021, 515, 512, 260
    img, want = HostImage(W, H), HostImage(W, H)
    descs, datas = [], []
    for ty in range(2):
        for tx in range(2):
            w = min(tw, W - tx * tw)
            h = min(th, H - ty * th)
            d, data, _, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1,
                                              tile=(tx * tw, ty * th, w, h), mcu=(2, 1),
                                              frame=(tw // 2, th))
            descs.append(d)
            datas.append(data)
    want_cons = []
    for d, data in zip(descs, datas):
        st, c = oracle.ljpeg(d, data, want)
        assert st == 0
        want_cons.append(c)
    rc, st, cons = gpu.dng_decompress_ljpeg(descs, datas, img.view())
    assert rc == 0 and not any(st)
    assert cons == want_cons
    assert np.array_equal(img.u16(), want.u16())


def _device_roundtrip(gpu, kind, jobs, blobs, out_bytes):
    import gpu_util
    inp = gpu_util.to_dev(np.concatenate(blobs))
    out = torch.zeros(out_bytes, dtype=torch.uint8, device="cuda")
    plan = gpu.ljpeg_plan(jobs) if kind == "ljpeg" else gpu.cr2_plan(jobs)
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st, cons = plan.results()
    assert rc == 0, (rc, st, gpu.last_error())
    return out.cpu().numpy(), cons


def test_cfg3_cr2_full_size_roundtrip(gpu):
    """BASELINE config 3: 6720x4480, 2 components, 3 slices (2x2240+2240)."""
    import gpu_util
    W, H = 6720, 4480
    src = synth.sensor_image(W, H, 14, seed=1)
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
    d.frame_w, d.frame_h = W // 2, H
    d.num_slices, d.slice_width, d.last_slice_width = 3, 2240, 2240
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, [2240, 2240, 2240])
    scan, bits = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [C.NIKON, C.NIKON])
    abi.fill_recipe(d, synth.huff_tables(C.NIKON), [0, 0], [1 << 13] * 2)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(30, np.uint8)])
    j = abi.Cr2Job()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
    j.img = gpu_util.image_job_view(W, H, 1, out_pitch(W, 1))
    got, cons = _device_roundtrip(gpu, "cr2", [j], [data], out_pitch(W, 1) * H)
    got = got.view(np.uint16).reshape(H, out_pitch(W, 1) // 2)[:, :W]
    assert cons[0] == len(scan)
    assert np.array_equal(got, src)


def test_cfg4_dng_tiles_full_size_roundtrip(gpu):
    """BASELINE config 4: 8192x5464 as 2x2 LJPEG tiles of 4096x2732, one plan."""
    import gpu_util
    W, H, tw, th = 8192, 5464, 4096, 2732
    src = synth.sensor_image(W, H, 14, seed=2)
    jobs, blobs, off, lens = [], [], 0, []
    for ty in range(2):
        for tx in range(2):
            tile = np.ascontiguousarray(src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw])
            scan, bits = synth.ljpeg_encode_scan(tile, 2, [1 << 13] * 2, [C.NIKON, C.NIKON])
            data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8),
                                   np.zeros(16 + (-(len(scan) + 18)) % 16, np.uint8)])
            d = abi.LJpegDesc()
            d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx * tw, ty * th, tw, th
            d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = 2, 1, tw // 2, th
            d.n_comp, d.rows_per_restart_interval = 2, th
            abi.fill_recipe(d, synth.huff_tables(C.NIKON), [0, 0], [1 << 13] * 2)
            j = abi.LJpegJob()
            j.desc = d
            j.in_offset, j.in_bytes, j.img_offset = off, data.size, 0
            j.img = gpu_util.image_job_view(W, H, 1, out_pitch(W, 1))
            jobs.append(j)
            blobs.append(data)
            lens.append(len(scan))
            off += data.size
    got, cons = _device_roundtrip(gpu, "ljpeg", jobs, blobs, out_pitch(W, 1) * H)
    got = got.view(np.uint16).reshape(H, out_pitch(W, 1) // 2)[:, :W]
    assert cons == lens
    assert np.array_equal(got, src)


def test_constant_regions_converge(gpu, oracle):
    """Blown highlights: inside a run of identical samples the bit stream is
    periodic and a mis-aligned speculative parse can cycle without ever meeting
    the true one (e.g. the Nikon table's SSSS=0 code '111110' repeated), so
    those subsequences only get their entry state by propagation.  Slow path,
    same result."""
    rng = np.random.default_rng(44)
    W, H = 2048, 96
    src = C.smooth_image(rng, H, W)
    src[10:60, 300:1900] = 16383           # saturated block, many subsequences long
    src[70:, :] = 0                        # black rows: whole workgroups of zeros
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, [W])
    scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [C.NIKON, C.NIKON])
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
    d.frame_w, d.frame_h = W // 2, H
    d.num_slices, d.slice_width, d.last_slice_width = 1, 0, W
    abi.fill_recipe(d, synth.huff_tables(C.NIKON), [0, 0], [1 << 13] * 2)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(16, np.uint8)])
    img, want = HostImage(W, H), HostImage(W, H)
    so = oracle.cr2(d, data, want)
    assert so[0] == 0 and gpu.cr2_decode(d, data, img.view()) == so
    assert np.array_equal(img.u16(), want.u16())
    assert np.array_equal(img.pixels(), src)


@pytest.mark.parametrize("value", [0, 16383, 8192])
def test_constant_frame(gpu, oracle, value):
    """A frame of one value (lens cap / fully blown): ~100 workgroups of a
    perfectly periodic bit stream.  No subsequence synchronises by itself; the
    transfer-function fallback gives every workgroup its entry state."""
    W, H = 4096, 512
    src = np.full((H, W), value, np.uint16)
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, [W])
    scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [C.NIKON, C.NIKON])
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
    d.frame_w, d.frame_h = W // 2, H
    d.num_slices, d.slice_width, d.last_slice_width = 1, 0, W
    abi.fill_recipe(d, synth.huff_tables(C.NIKON), [0, 0], [1 << 13] * 2)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(16, np.uint8)])
    img, want = HostImage(W, H), HostImage(W, H)
    so = oracle.cr2(d, data, want)
    assert so[0] == 0 and gpu.cr2_decode(d, data, img.view()) == so
    assert np.array_equal(img.u16(), want.u16())
    assert np.array_equal(img.pixels(), src)


# ---- Cr2sRawInterpolator --------------------------------------------------------

@pytest.mark.parametrize("c", G.SRAW_CASES, ids=lambda c: c["name"])
def test_sraw_interpolate_golden(gpu, oracle, c):
    d, px, (iw, ih), (ow, oh) = G.build_sraw(c)
    src = HostImage(iw, ih, 1, is_cfa=False)
    src.pixels()[:] = px
    got, want = HostImage(ow, oh, 3, is_cfa=False), HostImage(ow, oh, 3, is_cfa=False)
    assert gpu.sraw_interpolate(d, src.view(), got.view()) == oracle.sraw(d, src, want) == 0
    assert np.array_equal(got.u16(), want.u16())
    assert G.image_hash(got.pixels()) == GOLD["sraw"][c["name"]]["hash"]


def test_sraw_decode_then_interpolate_on_device(gpu, oracle):
    """The Cr2Decoder sRaw flow without leaving HBM: Cr2Decompressor <3,2,2> plan,
    then the interpolation plan on its output buffer."""
    import gpu_util
    rng = np.random.default_rng(77)
    d, data, src, _ = C.make_cr2_sraw_case(rng, 2, (3, 160, 128), 120)
    h, w = src.shape
    j = abi.Cr2Job()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
    sub_pitch = out_pitch(w, 1)
    j.img = gpu_util.image_job_view(w, h, 1, sub_pitch, is_cfa=False)
    d_in = gpu_util.to_dev(np.concatenate([data, np.zeros(64, np.uint8)]))
    d_sub = torch.zeros(sub_pitch * h, dtype=torch.uint8, device="cuda")
    p1 = gpu.cr2_plan([j])
    p1.run(d_in.data_ptr(), d_sub.data_ptr())
    assert p1.results()[:2] == (0, [0])
    sd = abi.SrawDesc.make(2, 2, [2000, 1024, 1500], -120)
    ow, oh = 2 * (w // 6), 2 * h
    want_sub, want = HostImage(w, h, 1, is_cfa=False), HostImage(ow, oh, 3, is_cfa=False)
    want_sub.pixels()[:] = src
    assert oracle.sraw(sd, want_sub, want) == 0
    sj = abi.SrawJob()
    sj.desc = sd
    sj.in_offset, sj.img_offset = 0, 0
    sj.in_ = gpu_util.image_job_view(w, h, 1, sub_pitch, is_cfa=False)
    sj.img = gpu_util.image_job_view(ow, oh, 3, want.pitch, is_cfa=False)
    d_out = torch.full((want.buf.size,), 0xA5, dtype=torch.uint8, device="cuda")
    p2 = gpu.sraw_plan([sj])
    p2.run(d_sub.data_ptr(), d_out.data_ptr())
    assert p2.results()[:2] == (0, [0])
    assert np.array_equal(d_out.cpu().numpy(), want.buf)
    p1.close()
    p2.close()


def test_host_pointer_calls_from_many_threads(gpu, oracle):
    """The host-pointer entry points stage on a lane of their own (buffers + stream):
    calls from different threads of one context run side by side and do not disturb
    each other (rstest-style file loops; the reference's tile threads)."""
    import threading
    rng = np.random.default_rng(77)
    work = []
    for k in range(8):
        W, H = 1024 + 64 * k, 160 + 8 * k
        d, data, _, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H),
                                          mcu=(2, 1))
        want = HostImage(W, H)
        so = oracle.ljpeg(d, data, want)
        work.append((d, data, W, H, want, so))
    # and some unpack work in between
    upk = []
    for k in range(4):
        W, H, bps = 2048, 64 + 16 * k, 12
        data = rng.integers(0, 256, size=H * W * bps // 8, dtype=np.uint8)
        d = abi.UnpackDesc(0, 0, W, H, W * bps // 8, bps, abi.ORDER_MSB)
        want = HostImage(W, H)
        assert oracle.unpack(d, data, want) == 0
        upk.append((d, data, W, H, want))
    errors = []

    def lj(item, reps):
        d, data, W, H, want, so = item
        for _ in range(reps):
            img = HostImage(W, H)
            got = gpu.ljpeg_decode(d, data, img.view())
            if got != so or not np.array_equal(img.u16(), want.u16()):
                errors.append(("ljpeg", W, H, got, so))

    def up(item, reps):
        d, data, W, H, want = item
        for _ in range(reps):
            img = HostImage(W, H)
            if gpu.unpack_u16(d, data, img.view()) != 0 or \
                    not np.array_equal(img.u16(), want.u16()):
                errors.append(("unpack", W, H))

    threads = [threading.Thread(target=lj, args=(w, 6)) for w in work] + \
              [threading.Thread(target=up, args=(u, 6)) for u in upk]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_mixed_plan_legacy_route_and_damaged_fused_stream(gpu, oracle):
    """One batched call with a 3-component tile (legacy route: int16 differences + K5 / K6)
    and a TRUNCATED 2-component tile (fused path -> flagged for the legacy route's
    end-of-stream rules).  The damaged stream's differences must not land in the healthy
    stream's scratch: the legacy kernels leave a fused-path stream alone until its own
    scratch exists."""
    rng = np.random.default_rng(77)
    W, H = 1280, 200
    dA, dataA, pxA, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, 768, H),
                                          mcu=(3, 1))
    dB, dataB, pxB, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(768, 0, 512, H),
                                          mcu=(2, 1))
    for cut in (len(dataB), len(dataB) // 3, len(dataB) // 2):
        bad = dataB[:cut]
        img, want = HostImage(W, H), HostImage(W, H)
        soA = oracle.ljpeg(dA, dataA, want)
        soB = oracle.ljpeg(dB, bad, want)
        rc, st, cons = gpu.dng_decompress_ljpeg([dA, dB], [dataA, bad], img.view())
        assert soA[0] == 0 and st[0] == 0 and cons[0] == soA[1]
        assert (st[1] != 0) == (soB[0] != 0), (cut, st, soB)
        assert np.array_equal(img.pixels()[:, :768], pxA), cut
        if soB[0] == 0:
            assert cons[1] == soB[1]
            assert np.array_equal(img.u16(), want.u16())


def test_concurrent_single_pass_kernels_from_host_threads(gpu, oracle):
    """The unbatched DNG / ArwDecoder OpenMP shape (AbstractDngDecompressor.cpp:240-252,
    ArwDecoder.cpp:371-404): several host threads, each decoding its own tile through
    rsx_ljpeg_decode at the same time -- several single-pass kernels with ticketed
    look-backs share the chip, each a hundred workgroups and more.  Pixels and consumed
    bytes of every call, and no call may give up on a look-back (status stays OK)."""
    import threading
    rng = np.random.default_rng(4711)
    work = []
    for k in range(6):
        W, H = 2048 + 256 * (k % 3), 768 + 64 * k
        d, data, _, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H),
                                          mcu=(2, 1))
        want = HostImage(W, H)
        so = oracle.ljpeg(d, data, want)
        assert so[0] == 0
        assert data.size > 90 * 255 * 64   # a hundred workgroups a call, give or take
        work.append((d, data, W, H, want, so))
    errors = []
    start = threading.Barrier(len(work))

    def lj(item):
        d, data, W, H, want, so = item
        start.wait()
        for rep in range(4):
            img = HostImage(W, H)
            got = gpu.ljpeg_decode(d, data, img.view())
            if got != so or not np.array_equal(img.u16(), want.u16()):
                errors.append((W, H, rep, got, so))

    threads = [threading.Thread(target=lj, args=(w,)) for w in work]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
