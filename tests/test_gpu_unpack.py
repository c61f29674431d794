"""GPU parity: the HIP unpack path (through the C-ABI) vs the oracle, the
golden hashes of the reference, and size-independent properties at the
BASELINE sizes."""
import json
import os

import numpy as np
import pytest
import torch

from rawspeed_amd import abi, synth

import golden_cases as G
from oracle_lib import HostImage, out_pitch

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "golden_hashes.json")) as f:
    GOLD = json.load(f)


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


@pytest.mark.parametrize("i", range(len(G.UNPACK_CASES)))
def test_unpack_golden_host_api(gpu, oracle, i):
    """rsx_unpack_u16 (host pointers, what the patched reference method calls)."""
    d, data, (w, h, cpp) = G.build_unpack(G.UNPACK_CASES[i])
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    v = img.view()
    st = gpu.unpack_u16(d, data, v)
    assert st == oracle.unpack(d, data, want) == GOLD["unpack"][str(i)]["status"]
    assert np.array_equal(img.u16(), want.u16())      # incl. untouched padding/rows
    assert G.image_hash(img.pixels()) == GOLD["unpack"][str(i)]["hash"]


def test_unpack_sweep_vs_oracle(gpu, oracle):
    """All orders x bps 1..16 x odd widths / paddings / row offsets, one plan
    per order mix (device-resident API)."""
    import gpu_util
    rng = np.random.default_rng(21)
    jobs, wants, metas = [], [], []
    in_chunks, in_off, out_off = [], 0, 0
    for order in range(4):
        for bps in range(1, 17):
            for w in (8, 24, 56, 1000, 8200):
                if (w * bps) % 8:
                    continue
                pad = int(rng.integers(0, 5))
                oy = int(rng.integers(0, 3))
                h = int(rng.integers(1, 5))
                pitch = w * bps // 8 + pad
                data = rng.integers(0, 256, size=h * pitch, dtype=np.uint8)
                d = abi.UnpackDesc(0, oy, w, h, pitch, bps, order)
                dim_y = h + oy - int(rng.integers(0, 2))  # sometimes clamps the last row
                dim_y = max(dim_y, oy)
                if dim_y == 0:
                    dim_y = 1
                want = HostImage(w, dim_y, 1)
                st = oracle.unpack(d, data, want)
                if st != 0:
                    # < 4 bytes of stream: the bit streamer refuses (BitStreamer.h:58-59)
                    assert st == abi.RSX_ERR_IO and h * pitch < 4
                    got = HostImage(w, dim_y, 1)
                    assert gpu.unpack_u16(d, data, got.view()) == st
                    continue
                j = abi.UnpackJob()
                j.desc = d
                j.in_offset, j.in_bytes = in_off, data.size
                j.img_offset = out_off
                j.img = gpu_util.image_job_view(w, dim_y, 1, want.pitch)
                jobs.append(j)
                wants.append(want)
                metas.append((order, bps, w, pad, oy, h, dim_y))
                in_chunks.append(data)
                # deliberately unaligned strip starts for a third of the jobs
                in_off += data.size + int(rng.integers(0, 3)) * 5
                in_chunks.append(np.zeros(in_off - sum(c.size for c in in_chunks), np.uint8))
                out_off += want.buf.size
    inp = gpu_util.to_dev(np.concatenate(in_chunks + [np.zeros(64, np.uint8)]))
    out = torch.full((out_off,), 0xA5, dtype=torch.uint8, device="cuda")
    plan = gpu.unpack_plan(jobs)
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st, _ = plan.results()
    assert rc == 0 and not any(st)
    got = out.cpu().numpy()
    for j, want, m in zip(jobs, wants, metas):
        g = got[j.img_offset:j.img_offset + want.buf.size]
        assert np.array_equal(g, want.buf), m


def test_unpack_16bit_lsb_honours_crop_x(gpu, oracle):
    # copyPixels fast path writes at offset.x (UncompressedDecompressor.cpp:257-264)
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, size=6 * 40, dtype=np.uint8)
    d = abi.UnpackDesc(3, 1, 20, 6, 40, 16, abi.ORDER_LSB)
    img, want = HostImage(24, 8), HostImage(24, 8)
    assert gpu.unpack_u16(d, data, img.view()) == oracle.unpack(d, data, want) == 0
    assert np.array_equal(img.u16(), want.u16())
    # ... while the packed paths ignore it (UncompressedDecompressor.cpp:196)
    data = rng.integers(0, 256, size=6 * 30, dtype=np.uint8)
    d = abi.UnpackDesc(3, 1, 20, 6, 30, 12, abi.ORDER_MSB)
    img, want = HostImage(24, 8), HostImage(24, 8)
    assert gpu.unpack_u16(d, data, img.view()) == oracle.unpack(d, data, want) == 0
    assert np.array_equal(img.u16(), want.u16())


def test_unpack_errors_match_oracle(gpu, oracle):
    data = np.zeros(64, np.uint8)
    for d in [abi.UnpackDesc(0, 0, 8, 40, 12, 12, 0), abi.UnpackDesc(0, 0, 0, 4, 12, 12, 0),
              abi.UnpackDesc(0, 0, 8, 4, 11, 12, 0), abi.UnpackDesc(0, 0, 8, 4, 12, 12, 4),
              abi.UnpackDesc(0, 0, 8, 4, 12, 17, 0), abi.UnpackDesc(0, 0, 7, 4, 12, 12, 0),
              abi.UnpackDesc(0, 5, 8, 4, 12, 12, 0), abi.UnpackDesc(1, 0, 8, 4, 12, 12, 0),
              abi.UnpackDesc(0, 0, 8, 1, 2, 2, 1)]:
        img, want = HostImage(8, 4), HostImage(8, 4)
        st = gpu.unpack_u16(d, data, img.view())
        assert st == oracle.unpack(d, data, want) != 0
        assert np.array_equal(img.buf, want.buf)  # nothing written


def test_dng_uncompressed_tiles(gpu, oracle):
    """AbstractDngDecompressor::decompressThread<1>: 2x2 tiles, 12-bit MSB (non
    8/16/32-bit integer => big-endian, AbstractDngDecompressor.cpp:64-77)."""
    rng = np.random.default_rng(4)
    W, H, tw, th, bps = 40, 13, 24, 8, 12
    img, want = HostImage(W, H), HostImage(W, H)
    descs, datas = [], []
    for ty in range(2):
        for tx in range(2):
            w = min(tw, W - tx * tw)
            h = min(th, H - ty * th)
            pitch = tw * bps // 8
            data = rng.integers(0, 256, size=th * pitch, dtype=np.uint8)
            d = abi.UnpackDesc(tx * tw, ty * th, w, h, pitch, bps, abi.ORDER_MSB)
            descs.append(d)
            datas.append(data)
            assert oracle.unpack(d, data, want) == 0
    rc, st = gpu.dng_decompress_uncompressed(descs, datas, img.view())
    assert rc == 0 and not any(st)
    assert np.array_equal(img.u16(), want.u16())


@pytest.mark.parametrize("cfg", ["cfg1_12bit_lsb_4096x3072", "cfg2_14bit_msb_8192x5464"])
def test_unpack_baseline_sizes_roundtrip(gpu, cfg):
    """BASELINE configs 1 and 2 at full size: pack(v) -> GPU unpack == v
    (size-independent round trip), plus a sampled oracle comparison."""
    import gpu_util
    from oracle_lib import Oracle
    if cfg.startswith("cfg1"):
        w, h, bps, order = 4096, 3072, 12, abi.ORDER_LSB
    else:
        w, h, bps, order = 8192, 5464, 14, abi.ORDER_MSB
    px = synth.uniform(w * h, bps, 42).reshape(h, w)
    packed = synth.pack_rows(px, bps, order)
    pitch = w * bps // 8
    opitch = out_pitch(w, 1)
    j = abi.UnpackJob()
    j.desc = abi.UnpackDesc(0, 0, w, h, pitch, bps, order)
    j.in_offset, j.in_bytes, j.img_offset = 0, packed.size, 0
    j.img = gpu_util.image_job_view(w, h, 1, opitch)
    inp = gpu_util.to_dev(packed)
    out = torch.zeros(opitch * h, dtype=torch.uint8, device="cuda")
    plan = gpu.unpack_plan([j])
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st, _ = plan.results()
    assert rc == 0
    got = out.cpu().numpy().view(np.uint16).reshape(h, opitch // 2)[:, :w]
    assert np.array_equal(got, px)
    # oracle on a 64-row strip from the middle
    r0 = h // 2
    d = abi.UnpackDesc(0, 0, w, 64, pitch, bps, order)
    want = HostImage(w, 64)
    assert Oracle().unpack(d, packed[r0 * pitch:(r0 + 64) * pitch], want) == 0
    assert np.array_equal(want.pixels(), got[r0:r0 + 64])


# ---- decode8BitRaw<true> / decode12BitRawWithControl<e> /
# ---- decode12BitRawUnpackedLeftAligned<e> ------------------------------------

@pytest.mark.parametrize("i", range(len(G.VARIANT_CASES)))
def test_unpack_variant_golden_host_api(gpu, oracle, i):
    d, data, (w, h, cpp) = G.build_variant(G.VARIANT_CASES[i])
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    st = gpu.unpack_variant_u16(d, data, img.view())
    assert st == oracle.unpack_variant(d, data, want) == GOLD["variant"][str(i)]["status"]
    assert np.array_equal(img.u16(), want.u16())
    assert G.image_hash(img.pixels()) == GOLD["variant"][str(i)]["hash"]


def test_unpack_variant_sweep_vs_oracle(gpu, oracle):
    """Every width class (mod 8, mod 10, segment boundaries of the kernels),
    aligned and unaligned image bases / pitches, one device-resident plan."""
    import gpu_util
    rng = np.random.default_rng(23)
    jobs, wants, in_chunks = [], [], []
    in_off = out_off = 0
    widths = list(range(2, 64, 2)) + [2558, 2560, 2562, 8192, 10246]
    for variant in range(3):
        for big in ((0,) if variant == 0 else (0, 1)):
            for w in widths:
                h = int(rng.integers(1, 4))
                bpl = G.variant_bpl(variant, w)
                data = rng.integers(0, 256, size=bpl * h + int(rng.integers(0, 3)),
                                    dtype=np.uint8)
                d = abi.UnpackVariantDesc(variant, big, w, h)
                dim_x = w + int(rng.integers(0, 3))
                dim_y = h + int(rng.integers(0, 2))
                want = HostImage(dim_x, dim_y, 1)
                assert oracle.unpack_variant(d, data, want) == 0
                j = abi.UnpackVariantJob()
                j.desc = d
                j.in_offset, j.in_bytes, j.img_offset = in_off, data.size, out_off
                j.img = gpu_util.image_job_view(dim_x, dim_y, 1, want.pitch)
                jobs.append(j)
                wants.append(want)
                in_chunks.append((in_off, data))
                in_off += data.size + int(rng.integers(0, 3)) * 3
                # a third of the images start 2 bytes off a 16-byte boundary
                out_off += want.buf.size + (2 if rng.integers(0, 3) == 0 else 0)
                out_off += (-out_off) % 2
    in_host = np.zeros(in_off + 16, dtype=np.uint8)
    for off, data in in_chunks:
        in_host[off:off + data.size] = data
    out_host = np.full(out_off + 16, 0xA5, dtype=np.uint8)
    d_in, d_out = gpu_util.to_dev(in_host), gpu_util.to_dev(out_host)
    plan = gpu.unpack_variant_plan(jobs)
    plan.run(d_in.data_ptr(), d_out.data_ptr())
    rc, status, _ = plan.results()
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert rc == 0 and all(s == 0 for s in status)
    for j, want in zip(jobs, wants):
        a = got[j.img_offset:j.img_offset + want.buf.size]
        assert np.array_equal(a, want.buf), (j.desc.variant, j.desc.big_endian, j.desc.w)
    plan.close()


def test_unpack_variant_errors(gpu, oracle):
    for variant, w, h, n in ((1, 11, 2, 64), (1, 10, 2, 31), (0, 10, 2, 19),
                             (2, 10, 2, 39), (0, 12, 2, 24), (4, 10, 2, 100)):
        d = abi.UnpackVariantDesc(variant, 0, w, h)
        data = np.zeros(n, dtype=np.uint8)
        img, want = HostImage(10, 2, 1), HostImage(10, 2, 1)
        st = gpu.unpack_variant_u16(d, data, img.view())
        assert st != 0 and st == oracle.unpack_variant(d, data, want)
        assert np.array_equal(img.u16(), want.u16())


def test_unpack_variant_full_frame_properties(gpu):
    """Full sensor size: the with-control layout is the 12-bit packed layout
    with one byte inserted every 15; both kernels must agree, and the
    left-aligned variant is the 16-bit unpack shifted by 4."""
    import gpu_util
    w, h = 8280, 5520  # multiple of 10: every unit is complete
    rng = np.random.default_rng(3)
    pix = rng.integers(0, 4096, size=(h, w), dtype=np.uint16)
    packed = synth.pack_rows(pix, 12, abi.ORDER_MSB)          # (h, w*12/8)
    units = packed.reshape(h, w // 10, 15)
    ctrl = np.concatenate([units, np.full((h, w // 10, 1), 0x5A, np.uint8)], axis=2)
    ctrl = np.ascontiguousarray(ctrl.reshape(h, -1))
    assert ctrl.shape[1] == G.variant_bpl(1, w)
    pitch = out_pitch(w, 1)
    j = abi.UnpackVariantJob()
    j.desc = abi.UnpackVariantDesc(abi.UNPACK_12BIT_WITH_CONTROL, 1, w, h)
    j.in_offset, j.in_bytes, j.img_offset = 0, ctrl.size, 0
    j.img = gpu_util.image_job_view(w, h, 1, pitch)
    d_in = gpu_util.to_dev(ctrl.reshape(-1))
    d_out = torch.zeros(pitch * h, dtype=torch.uint8, device="cuda")
    plan = gpu.unpack_variant_plan([j])
    plan.run(d_in.data_ptr(), d_out.data_ptr())
    assert plan.results()[:2] == (0, [0])
    got = d_out.cpu().numpy().view(np.uint16).reshape(h, pitch // 2)[:, :w]
    assert np.array_equal(got, pix)
    plan.close()

    left = (pix.astype(np.uint16) << 4) | rng.integers(0, 16, size=pix.shape, dtype=np.uint16)
    for big in (0, 1):
        raw = left.astype(">u2" if big else "<u2").view(np.uint8).reshape(-1)
        j.desc = abi.UnpackVariantDesc(abi.UNPACK_12BIT_UNPACKED_LEFT_ALIGNED, big, w, h)
        j.in_bytes = raw.size
        d_in = gpu_util.to_dev(raw)
        d_out.zero_()
        plan = gpu.unpack_variant_plan([j])
        plan.run(d_in.data_ptr(), d_out.data_ptr())
        assert plan.results()[:2] == (0, [0])
        got = d_out.cpu().numpy().view(np.uint16).reshape(h, pitch // 2)[:, :w]
        assert np.array_equal(got, pix)
        plan.close()


# ---- RawImageType::F32 images (decodePackedFP fp16 / fp24, 32-bit copy) -------

@pytest.mark.parametrize("i", range(len(G.F32_CASES)))
def test_unpack_f32_golden_host_api(gpu, oracle, i):
    d, data, (w, h, cpp) = G.build_f32(G.F32_CASES[i])
    img, want = HostImage(w, h, cpp, bpc=4), HostImage(w, h, cpp, bpc=4)
    st = gpu.unpack_f32(d, data, img.view())
    assert st == oracle.unpack_f32(d, data, want) == GOLD["f32"][str(i)]["status"] == 0
    assert np.array_equal(img.u32(), want.u32())
    assert G.image_hash(img.u32()[:, :w * cpp]) == GOLD["f32"][str(i)]["hash"]


def test_unpack_f32_every_half_and_sweep(gpu, oracle):
    """All 65536 binary16 patterns in both byte orders, then a shape sweep of
    fp24 / fp16 / 32-bit jobs in one device-resident plan."""
    import gpu_util
    allh = np.arange(65536, dtype=np.uint16)
    for order in (0, 1):
        data = allh.astype(">u2" if order else "<u2").view(np.uint8)
        d = abi.UnpackDesc(0, 0, 1024, 64, 2048, 16, order)
        img, want = HostImage(1024, 64, 1, bpc=4), HostImage(1024, 64, 1, bpc=4)
        assert gpu.unpack_f32(d, data, img.view()) == oracle.unpack_f32(d, data, want) == 0
        assert np.array_equal(img.u32(), want.u32())
        f = img.u32()[:, :1024].reshape(-1).view(np.float32)
        ref = allh.view(np.float16).astype(np.float32)
        ok = np.isnan(ref) | (f == ref)
        assert ok.all()                     # numpy agrees wherever it is a number

    rng = np.random.default_rng(24)
    jobs, wants, chunks = [], [], []
    in_off = out_off = 0
    for order, bps in ((0, 16), (1, 16), (0, 24), (1, 24), (0, 32), (3, 32)):
        for w in (1, 2, 3, 5, 64, 1023, 1024, 1025, 4100):
            cpp = int(rng.integers(1, 4))
            if (w * cpp * bps) % 8:
                continue
            h, pad = int(rng.integers(1, 4)), int(rng.integers(0, 4))
            ox, oy = int(rng.integers(0, 3)), int(rng.integers(0, 2))
            pitch = w * cpp * bps // 8 + pad
            if h * pitch < 4:
                continue
            data = rng.integers(0, 256, size=h * pitch, dtype=np.uint8)
            d = abi.UnpackDesc(ox, oy, w, h, pitch, bps, order)
            want = HostImage(w + ox, h + oy, cpp, bpc=4)
            assert oracle.unpack_f32(d, data, want) == 0
            j = abi.UnpackJob()
            j.desc = d
            j.in_offset, j.in_bytes, j.img_offset = in_off, data.size, out_off
            j.img = gpu_util.image_job_view(w + ox, h + oy, cpp, want.pitch)
            jobs.append(j)
            wants.append(want)
            chunks.append((in_off, data))
            in_off += data.size + int(rng.integers(0, 3)) * 7
            out_off += want.buf.size
    in_host = np.zeros(in_off + 16, np.uint8)
    for off, data in chunks:
        in_host[off:off + data.size] = data
    d_in = gpu_util.to_dev(in_host)
    d_out = torch.full((out_off + 16,), 0xA5, dtype=torch.uint8, device="cuda")
    plan = gpu.unpack_f32_plan(jobs)
    plan.run(d_in.data_ptr(), d_out.data_ptr())
    rc, status, _ = plan.results()
    assert rc == 0 and all(s == 0 for s in status)
    got = d_out.cpu().numpy()
    for j, want in zip(jobs, wants):
        assert np.array_equal(got[j.img_offset:j.img_offset + want.buf.size], want.buf), \
            (j.desc.bit_order, j.desc.bits_per_pixel, j.desc.crop_w)
    plan.close()


def test_decode8bit_lookup(gpu, oracle):
    """decode8BitRaw<false>: several frames with different tables in one plan."""
    import gpu_util
    from oracle_lib import dither_lut8
    rng = np.random.default_rng(25)
    jobs, wants, chunks = [], [], []
    in_off = out_off = 0
    for w, h in ((16, 3), (250, 4), (8200, 2), (4096, 5)):
        curve = np.sort(rng.integers(0, 65536, size=256)).astype(np.uint16)
        d = abi.UnpackVariantDesc(abi.UNPACK_8BIT_LOOKUP, 0, w, h).set_lut(dither_lut8(curve))
        data = rng.integers(0, 256, size=w * h, dtype=np.uint8)
        want, img = HostImage(w, h, 1), HostImage(w, h, 1)
        assert oracle.unpack_variant(d, data, want) == 0
        assert gpu.unpack_variant_u16(d, data, img.view()) == 0      # host-pointer call
        assert np.array_equal(img.u16(), want.u16())
        j = abi.UnpackVariantJob()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = in_off, data.size, out_off
        j.img = gpu_util.image_job_view(w, h, 1, want.pitch)
        jobs.append(j)
        wants.append(want)
        chunks.append((in_off, data))
        in_off += data.size + 5
        out_off += want.buf.size
    in_host = np.zeros(in_off + 16, np.uint8)
    for off, data in chunks:
        in_host[off:off + data.size] = data
    d_in = gpu_util.to_dev(in_host)
    d_out = torch.full((out_off + 16,), 0xA5, dtype=torch.uint8, device="cuda")
    plan = gpu.unpack_variant_plan(jobs)
    plan.run(d_in.data_ptr(), d_out.data_ptr())
    rc, status, _ = plan.results()
    assert rc == 0 and status == [0] * len(jobs)
    got = d_out.cpu().numpy()
    for j, want in zip(jobs, wants):
        assert np.array_equal(got[j.img_offset:j.img_offset + want.buf.size], want.buf)
    plan.close()


@pytest.mark.parametrize("bps,order,crop", [(14, abi.ORDER_MSB, (0, 0)), (12, abi.ORDER_LSB, (16, 7)),
                                            (16, abi.ORDER_LSB, (8, 3)), (10, abi.ORDER_MSB, (0, 5))])
def test_host_pointer_call_in_bands(gpu, bps, order, crop, monkeypatch):
    """Host pointers and a large input: rsx_unpack_u16 uploads, unpacks and downloads in row
    bands (a helper thread uploads band k + 1 under the download of band k, rsx_api.hip
    unpack_host).  Same pixels as the one-shot path and as the size-independent round trip
    pack(v) -> unpack == v, with a crop offset, and nothing outside the rectangle touched."""
    w, h = 6144, 3000
    cx, cy = crop
    px = synth.uniform(w * h, bps, 7).reshape(h, w)
    packed = synth.pack_rows(px, bps, order)
    pitch = w * bps // 8
    assert packed.size >= 16 << 20
    d = abi.UnpackDesc(cx, cy, w, h, pitch, bps, order)
    img = HostImage(w + cx + 5, h + cy + 2)
    assert gpu.unpack_u16(d, packed, img.view()) == 0
    got = img.u16()
    # (only the 16-bit little-endian copy honours the x offset; the packed walks start their
    # rows at column 0, UncompressedDecompressor.cpp:196)
    x0 = cx if (bps == 16 and order == abi.ORDER_LSB) else 0
    assert np.array_equal(got[cy:cy + h, x0:x0 + w], px)
    # the same call in one piece (the switch is read when a context is created)
    from rawspeed_amd import capi
    ref = HostImage(w + cx + 5, h + cy + 2)
    monkeypatch.setenv("RSX_HOST_NO_OVERLAP", "1")
    one_piece = capi.Context(0)
    assert one_piece.unpack_u16(d, packed, ref.view()) == 0
    assert np.array_equal(img.buf, ref.buf)


def test_host_pointer_tiles_in_bands(gpu, oracle):
    """Several large tiles in one call (an uncompressed DNG through
    AbstractDngDecompressor::decompressThread<1>): every tile is one or more bands; tiles of
    different bit orders, a clipped right column of tiles."""
    rng = np.random.default_rng(11)
    W, H, tw, th = 6000, 4096, 2048, 2048
    img, want = HostImage(W, H), HostImage(W, H)
    descs, datas = [], []
    for ty in range(2):
        for tx in range(3):
            bps, order = (12, abi.ORDER_MSB) if (tx + ty) % 2 == 0 else (16, abi.ORDER_LSB)
            w = min(tw, W - tx * tw)
            pitch = tw * bps // 8
            data = rng.integers(0, 256, size=th * pitch, dtype=np.uint8)
            d = abi.UnpackDesc(tx * tw, ty * th, w, th, pitch, bps, order)
            descs.append(d)
            datas.append(data)
            assert oracle.unpack(d, data, want) == 0
    assert sum(x.size for x in datas) >= 16 << 20
    rc, st = gpu.dng_decompress_uncompressed(descs, datas, img.view())
    assert rc == 0 and not any(st)
    assert np.array_equal(img.u16(), want.u16())
