"""The drop-in boundary exercised for real: the reference's own host code
(LJpegDecoder / Cr2LJpegDecoder header parsing, AbstractDngDecompressor tile
fan-out with OpenMP, UncompressedDecompressor) built twice from /root/reference --
unmodified (oracle/_ref/librawspeed_ref.so) and with INTEGRATION.md's three
forwarding hunks applied and linked against librsx.so
(oracle/_ref/librawspeed_rsx.so, recipe: oracle/make_patched.py + Makefile).
The same entry points must give bit-identical RawImages."""
import numpy as np
import pytest

from rawspeed_amd import abi, capi, synth

import cases as C
from oracle_lib import REF_RSX_SO, Ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    if not (Ref.available() and Ref.available(REF_RSX_SO)):
        pytest.skip("oracle/_ref builds absent")
    capi.lib()  # torch's HIP runtime first (one runtime per process)
    return Ref(), Ref(REF_RSX_SO)


class Forwarding:
    """What the patched build's hunks did with the units of work of a call (a strip, a
    scan, a DNG tile): `fwd` decoded by the device, `fell` left to the method's original
    body (rsx_shim::stats()).  The hunks fall through on ANY non-OK status, so an
    identical image alone proves nothing: every test states what it expects here."""

    def __init__(self, rsx):
        self.rsx = rsx
        self.c0 = rsx.rsx_counts()

    def delta(self):
        c1 = self.rsx.rsx_counts()
        return c1[1] - self.c0[1], c1[2] - self.c0[2]


def both(pair, fn, dims, fwd=1, fell=0):
    """Run fn on the unmodified and on the patched build.  fwd / fell: units the patched
    build must have forwarded to the device / left to the CPU loop (None = do not check;
    a (lo, hi) tuple = a range)."""
    out = []
    for k, lib in enumerate(pair):
        img = lib.image(*dims)
        f = Forwarding(lib) if k == 1 else None
        st = fn(lib, img)
        if f is not None:
            if fwd == "auto":  # the unmodified build decides: success = forwarded, failure = CPU loop
                st0 = out[0][0]
                ok = (st0[0] if isinstance(st0, tuple) else st0) == 0
                # (a failure the reference detects before its hot-path method never reaches
                # the hunk: then nothing is counted at all)
                check_forwarding(f, 1 if ok else 0, 0 if ok else (0, 1))
            else:
                check_forwarding(f, fwd, fell)
        out.append((st, img.u16().copy(), lib.last_error()))
    return out


def check_forwarding(f, fwd=1, fell=0):
    got_fwd, got_fell = f.delta()
    for name, want, got in (("forwarded", fwd, got_fwd), ("fell through", fell, got_fell)):
        if want is None:
            continue
        lo, hi = want if isinstance(want, tuple) else (want, want)
        assert lo <= got <= hi, "%s: %d units, expected %s (forwarded %d, fell through %d)" % (
            name, got, want, got_fwd, got_fell)


def test_ljpeg_container_full_image(pair):
    rng = np.random.default_rng(31)
    W, H = 1536, 400
    src = C.smooth_image(rng, H, W)
    blob, _, _, _ = synth.ljpeg_container(src, 2, 14, [0, 0], [C.NIKON])
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.ljpeg_container(blob, img, 0, 0, W, H, (W, H)), (W, H, 1))
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)
    assert np.array_equal(a[:, :W], src)


def test_ljpeg_container_two_tables_and_dri(pair):
    rng = np.random.default_rng(32)
    W, H = 1024, 240
    src = C.smooth_image(rng, H, W)
    blob, _, _, _ = synth.ljpeg_container(src, 2, 14, [0, 1], [C.NIKON, C.ALT], rows_per_ri=40)
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.ljpeg_container(blob, img, 0, 0, W, H, (W, H)), (W, H, 1))
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)


def test_cr2_container_slices(pair):
    rng = np.random.default_rng(33)
    W, H = 2016, 1100   # frame.w*cps = 2016 <= 2*1100: no Canon double-height quirk
    src = C.smooth_image(rng, H, W)
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, [672, 672, 672])
    blob, _, _, _ = synth.ljpeg_container(rows, 2, 14, [0, 0], [C.NIKON])
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.cr2_container(blob, img, 3, 672, 672), (W, H, 1))
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)
    assert np.array_equal(a[:, :W], src)


@pytest.mark.parametrize("ysf", [1, 2])
def test_cr2_sraw_container(pair, ysf):
    """Canon sRaw: Cr2LJpegDecoder reads the 2x1 / 2x2 luma sampling factors
    from the SOF, fixes up the <3,2,1> slice widths (Cr2LJpegDecoder.cpp:88-120)
    and hands <3,2,ysf> to Cr2Decompressor -> forwarding hunk -> librsx."""
    rng = np.random.default_rng([35, ysf])
    groups = (3, 216, 160)
    d, data, src, _, rows = C.make_cr2_sraw_case(rng, ysf, groups, 300, with_rows=True)
    blob, _, _, _ = synth.ljpeg_container(
        rows, 3, 14, [0, 0, 0], [C.NIKON], frame_wh=(d.frame_w, d.frame_h),
        samp=[(2, ysf), (1, 1), (1, 1)], pattern=synth.SRAW_PATTERN[2 + 2 * ysf])
    # <3,2,1>: the decoder multiplies the slice widths by 3/2 itself
    unit = 4 if ysf == 1 else 6
    h, w = src.shape
    def run(lib, img):
        img.set_subsampling(2, ysf)  # what Cr2Decoder sets for sRaw files
        return lib.cr2_container(blob, img, 3, groups[1] * unit, groups[2] * unit)

    (s0, a, e0), (s1, b, e1) = both(pair, run, (w, h, 1, False))
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)
    assert np.array_equal(a[:, :w], src)


@pytest.mark.parametrize("threads", [1, 4])
def test_dng_ljpeg_tiles_through_reference_fanout(pair, threads):
    """AbstractDngDecompressor::decompress(): odd-sized image, 2x3 tiles; with 4
    OpenMP threads the tile threads enter the C-ABI concurrently."""
    rng = np.random.default_rng(34)
    W, H, tw, th = 1021, 700, 512, 256
    src = C.smooth_image(rng, H, W)
    blobs = []
    for ty in range((H + th - 1) // th):
        for tx in range((W + tw - 1) // tw):
            tile = np.zeros((th, tw), np.uint16)
            part = src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            tile[:part.shape[0], :part.shape[1]] = part
            tile[part.shape[0]:, :] = 1000
            tile[:, part.shape[1]:] = 1000
            blob, _, _, _ = synth.ljpeg_container(tile, 2, 14, [0, 0], [C.NIKON])
            blobs.append(blob)
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.dng(img, 7, tw, th, blobs, threads=threads), (W, H, 1),
        fwd=len(blobs))
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)
    assert np.array_equal(a[:, :W], src)


def test_dng_uncompressed_tiles_through_reference_fanout(pair):
    """decompressThread<1>.  Packed (non 8/16/32-bit) tiles write from column 0
    whatever their x offset (UncompressedDecompressor.cpp:196), so with several
    tile COLUMNS the reference's own OpenMP threads race on the same pixels; the
    comparable shapes are one tile column of packed data, and multi-column 16-bit
    tiles (the copyPixels path honours offset.x)."""
    rng = np.random.default_rng(35)
    # one column of 12-bit tiles, 3 tile rows, bottom one overhanging
    W, H, tw, th, bps = 256, 300, 256, 128, 12
    blobs = [rng.integers(0, 256, size=th * tw * bps // 8, dtype=np.uint8)
             for _ in range((H + th - 1) // th)]
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.dng(img, 1, tw, th, blobs, bps=bps, threads=4), (W, H, 1),
        fwd=len(blobs))
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)
    # 4 x 3 tiles of 16-bit little-endian data
    W, H, tw, th, bps = 1000, 300, 256, 128, 16
    blobs = [rng.integers(0, 256, size=th * tw * bps // 8, dtype=np.uint8)
             for _ in range(((H + th - 1) // th) * ((W + tw - 1) // tw))]
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.dng(img, 1, tw, th, blobs, bps=bps, threads=4), (W, H, 1),
        fwd=len(blobs))
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)


def test_unpack_entry_point(pair):
    rng = np.random.default_rng(36)
    for order, bps in ((abi.ORDER_LSB, 12), (abi.ORDER_MSB, 14), (abi.ORDER_MSB16, 10),
                       (abi.ORDER_MSB32, 12), (abi.ORDER_LSB, 16)):
        W, H = 2048, 64
        pitch = W * bps // 8 + 4
        data = rng.integers(0, 256, size=H * pitch, dtype=np.uint8)
        d = abi.UnpackDesc(0, 0, W, H, pitch, bps, order)
        (s0, a, _), (s1, b, _) = both(pair, lambda lib, img: lib.unpack(d, data, img), (W, H, 1))
        assert s0 == 0 and s1 == 0
        assert np.array_equal(a, b), (order, bps)


def test_corrupt_tile_is_reported_by_both(pair):
    rng = np.random.default_rng(37)
    W, H = 512, 128
    src = C.smooth_image(rng, H, W)
    blob, hdr, scan_len, _ = synth.ljpeg_container(src, 2, 14, [0, 0], [C.NIKON])
    bad = blob.copy()
    bad[hdr + 100:hdr + 102] = 0xFF   # FF FF: the scan ends here, far too early
    res = both(pair, lambda lib, img: lib.ljpeg_container(bad, img, 0, 0, W, H, (W, H)),
               (W, H, 1), fwd=0, fell=1)
    assert res[0][0] != 0 and res[1][0] != 0


def test_uncompressed_variant_methods(pair):
    """decode8BitRaw<true>, decode12BitRawWithControl<e> and
    decode12BitRawUnpackedLeftAligned<e> through the patched reference class
    (constructed the way DcsDecoder / ErfDecoder / OrfDecoder / Rw2Decoder do)."""
    rng = np.random.default_rng(36)
    for variant in range(3):
        for big in ((0,) if variant == 0 else (0, 1)):
            for (w, h, cut) in ((3000, 37, 0), (126, 5, 0), (500, 4, 9)):
                bpl = (w, 12 * w // 8 + (w + 2) // 10, 2 * w)[variant]
                data = rng.integers(0, 256, size=bpl * h - cut, dtype=np.uint8)
                d = abi.UnpackVariantDesc(variant, big, w, h)
                (s0, a, e0), (s1, b, e1) = both(
                    pair, lambda lib, img: lib.unpack_variant(d, data, img), (w, h, 1), fwd="auto")
                assert s0 == s1 and (s0 == 0) == (cut == 0), (variant, big, w, e0, e1)
                assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["lossless14_dither", "lossy12_split", "lossy14_z7_split",
                                  "literal_curve", "medium_image", "medium_split"])
def test_nikon_decompressor(pair, name):
    """NikonDecompressor: the reference constructor parses the makernote blob
    (curve, split, pUp), the patched decompress() forwards to librsx."""
    import golden_cases as G
    c = next(c for c in G.NIKON_CASES if c["name"] == name)
    meta, d, data, (w, h, cpp), _ = G.build_nikon(c)
    for unc in (True, False):
        (s0, a, e0), (s1, b, e1) = both(
            pair, lambda lib, img: lib.nikon(meta, c["bits"], data, img, unc), (w, h, cpp))
        assert s0 == 0 and s1 == 0, (e0, e1)
        assert np.array_equal(a, b)


def test_decode8bit_lookup_method(pair):
    """decode8BitRaw<false> under a RawImageCurveGuard (DcsDecoder.cpp:68-81)."""
    rng = np.random.default_rng(37)
    curve = np.sort(rng.integers(0, 65536, size=256)).astype(np.uint16)
    w, h = 3000, 41
    data = rng.integers(0, 256, size=w * h, dtype=np.uint8)
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.decode8bit_lookup(curve, w, h, data, img), (w, h, 1))
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)


def test_uncompressed_f32_image(pair):
    """readUncompressedRaw on a RawImageType::F32 image: fp16 / fp24 widening and
    the 32-bit copy, through the patched reference method."""
    rng = np.random.default_rng(38)
    for order, bps in ((1, 16), (0, 24), (0, 32)):
        w, h, cpp = 1200, 33, 3
        pitch = w * cpp * bps // 8 + 4
        data = rng.integers(0, 256, size=h * pitch, dtype=np.uint8)
        d = abi.UnpackDesc(2, 1, w, h, pitch, bps, order)
        out = []
        for lib in pair:
            img = lib.image(w + 2, h + 1, cpp, f32=True)
            f = Forwarding(lib)
            out.append((lib.unpack(d, data, img), img.u32().copy(), lib.last_error()))
            if lib is pair[1]:
                check_forwarding(f, 1, 0)
        (s0, a, e0), (s1, b, e1) = out
        assert s0 == 0 and s1 == 0, (e0, e1)
        assert np.array_equal(a, b)


def test_pentax_decompressor(pair):
    """PentaxDecompressor: the constructor rebuilds the Huffman table from the
    makernote blob, the patched decompress() forwards it to librsx."""
    import golden_cases as G
    for name in ("legacy_medium", "modern_wide", "range_error"):
        c = next(c for c in G.PENTAX_CASES if c["name"] == name)
        meta, d, data, (w, h, cpp), _ = G.build_pentax(c)
        (s0, a, e0), (s1, b, e1) = both(
            pair, lambda lib, img: lib.pentax(meta, data, img), (w, h, cpp), fwd="auto")
        assert s0 == s1, (e0, e1)
        if s0 == 0:
            assert np.array_equal(a, b)


def test_samsung_v1_decompressor(pair):
    import golden_cases as G
    for name in ("medium", "max_width", "range_error"):
        c = next(c for c in G.SAMSUNG_V1_CASES if c["name"] == name)
        d, data, (w, h, cpp), _ = G.build_samsung_v1(c)
        (s0, a, e0), (s1, b, e1) = both(
            pair, lambda lib, img: lib.samsung_v1(12, data, img), (w, h, cpp), fwd="auto")
        assert s0 == s1, (e0, e1)
        if s0 == 0:
            assert np.array_equal(a, b)


def test_samsung_v2_decompressor(pair):
    """SamsungV2Decompressor::decompress() through the patched class: streams of every
    optimisation-flag combination (forwarded, same image) and damaged ones (the original
    body takes over: same exception)."""
    import samsung_v2_cases as V2
    from test_oracle_samsung_v2 import _target
    if not hasattr(pair[0].lib, "ref_samsung_v2_decompress"):
        pytest.skip("oracle/_ref predates the SamsungV2 entry point")
    rng = np.random.default_rng(77)
    for k in range(10):
        bits = (12, 14)[k & 1]
        h, w = int(rng.integers(4, 60)), 16 * int(rng.integers(2, 24))
        data, want = V2.encode(rng, _target(rng, h, w, bits), bits, k % 8)
        if k >= 8:
            data = data[:16 + (data.size - 16) // 2]
        (s0, a, e0), (s1, b, e1) = both(
            pair, lambda lib, img: lib.samsung_v2(bits, data, img), (w, h, 1),
            fwd=1 if k < 8 else 0, fell=0 if k < 8 else 1)
        assert s0 == s1, (e0, e1)
        assert (s0 == 0) == (k < 8)
        if s0 == 0:
            assert np.array_equal(a, b) and np.array_equal(a, want)
        else:
            assert e0 == e1


def test_sraw_interpolator(pair):
    """Cr2sRawInterpolator::interpolate through the patched class."""
    import golden_cases as G
    for name in ("422_v0", "422_v2_medium", "420_v1_medium", "420_v2_extreme"):
        c = next(c for c in G.SRAW_CASES if c["name"] == name)
        d, px, (iw, ih), (ow, oh) = G.build_sraw(c)
        out = []
        for lib in pair:
            src, dst = lib.image(iw, ih, 1, False), lib.image(ow, oh, 3, False)
            src.set_pixels(px)
            f = Forwarding(lib)
            out.append((lib.sraw(d, src, dst), dst.u16().copy(), lib.last_error()))
            if lib is pair[1]:
                check_forwarding(f, 1, 0)
        (s0, a, e0), (s1, b, e1) = out
        assert s0 == 0 and s1 == 0, (e0, e1)
        assert np.array_equal(a, b)


def test_hasselblad_decompressor(pair):
    import golden_cases as G
    for name in ("full_range", "medium", "max_width"):
        c = next(c for c in G.HASSELBLAD_CASES if c["name"] == name)
        d, data, (w, h, cpp), _ = G.build_hasselblad(c)
        (s0, a, e0), (s1, b, e1) = both(
            pair, lambda lib, img: lib.hasselblad(d, data, img), (w, h, cpp))
        assert s0 == s1 and s0[0] == 0, (s0, s1, e0, e1)
        assert np.array_equal(a, b)


def test_sony_arw1_decompressor(pair):
    import golden_cases as G
    for name in ("medium_odd_width", "tall", "wide", "range_error"):
        c = next(c for c in G.SONY_ARW1_CASES if c["name"] == name)
        data, (w, h, cpp), _ = G.build_sony_arw1(c)
        (s0, a, e0), (s1, b, e1) = both(
            pair, lambda lib, img: lib.sony_arw1(data, img), (w, h, cpp), fwd="auto")
        assert s0 == s1, (e0, e1)
        if s0 == 0:
            assert np.array_equal(a, b)


# ---- one batched call per DNG, fall-through, error parity ---------------------------
def _dng_case(rng, W, H, tw, th, corrupt=()):
    src = C.smooth_image(rng, H, W)
    blobs = []
    k = 0
    for ty in range((H + th - 1) // th):
        for tx in range((W + tw - 1) // tw):
            tile = np.full((th, tw), 1000, np.uint16)
            part = src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            tile[:part.shape[0], :part.shape[1]] = part
            blob, hdr, scan_len, _ = synth.ljpeg_container(tile, 2, 14, [0, 0], [C.NIKON])
            if k in corrupt:
                blob = blob.copy()
                blob[hdr + scan_len // 2:hdr + scan_len // 2 + 2] = 0xFF  # FF FF: early end
            blobs.append(blob)
            k += 1
    return src, blobs


@pytest.mark.parametrize("threads", [1, 4])
def test_dng_decompress_makes_one_batched_call(pair, threads):
    """AbstractDngDecompressor::decompress() of the patched build: the reference's own
    tile fan-out runs (headers, marker walks), the tiles are decoded by ONE
    rsx_dng_decompress_ljpeg call."""
    ref, rsx = pair
    rng = np.random.default_rng(41)
    W, H, tw, th = 1500, 700, 512, 256   # 3 x 3 tiles, right and bottom ones overhang
    src, blobs = _dng_case(rng, W, H, tw, th)
    before = rsx.rsx_host_calls()
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.dng(img, 7, tw, th, blobs, threads=threads), (W, H, 1),
        fwd=len(blobs))
    assert ref.rsx_host_calls() == -1
    assert rsx.rsx_host_calls() - before == 1
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b) and np.array_equal(a[:, :W], src)


@pytest.mark.parametrize("threads", [1, 4])
def test_dng_uncompressed_tiles_make_one_batched_call(pair, threads):
    ref, rsx = pair
    rng = np.random.default_rng(42)
    W, H, tw, th, bps = 1000, 300, 256, 128, 16
    blobs = [rng.integers(0, 256, size=th * tw * bps // 8, dtype=np.uint8)
             for _ in range(((H + th - 1) // th) * ((W + tw - 1) // tw))]
    before = rsx.rsx_host_calls()
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.dng(img, 1, tw, th, blobs, bps=bps, threads=threads),
        (W, H, 1), fwd=len(blobs))
    assert rsx.rsx_host_calls() - before == 1
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("corrupt", [(4,), (0, 7)])
def test_dng_corrupt_tiles_same_image_and_error_log(pair, corrupt):
    """A damaged tile: the device path reports it, the tile is redone by the original
    CPU loop -- same partial image, same ErrorLog, same "too many errors" decision."""
    rng = np.random.default_rng(43)
    W, H, tw, th = 1536, 768, 512, 256
    src, blobs = _dng_case(rng, W, H, tw, th, corrupt)
    out = []
    for lib in pair:
        img = lib.image(W, H, 1)
        f = Forwarding(lib)
        st = lib.dng(img, 7, tw, th, blobs, threads=2)
        if lib is pair[1]:  # the good tiles from the device, the damaged ones from the CPU loop
            check_forwarding(f, len(blobs) - len(corrupt), (len(corrupt), 2 * len(corrupt)))
        out.append((st, img.u16().copy(), lib.last_error(), lib.image_errors(img)))
    (s0, a, e0, log0), (s1, b, e1, log1) = out
    assert s0 == s1 != 0  # (isTooManyErrors(1): one failed tile fails the image)
    assert e0 == e1
    assert sorted(log0.splitlines()) == sorted(log1.splitlines()) and log0
    assert np.array_equal(a, b)


def test_corrupt_scan_same_partial_image_and_message(pair):
    """LJpegDecoder on a stream that ends early: the patched build falls through to the
    reference's own loop, so the exception text and the partially decoded image agree."""
    rng = np.random.default_rng(44)
    W, H = 512, 128
    src = C.smooth_image(rng, H, W)
    blob, hdr, scan_len, _ = synth.ljpeg_container(src, 2, 14, [0, 0], [C.NIKON])
    bad = blob.copy()
    bad[hdr + scan_len // 2:hdr + scan_len // 2 + 2] = 0xFF
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.ljpeg_container(bad, img, 0, 0, W, H, (W, H)), (W, H, 1),
        fwd=0, fell=1)
    assert s0 == s1 != 0 and e0 == e1 and e0
    assert np.array_equal(a, b)


def test_unsupported_shape_falls_through_to_the_cpu_loop(pair):
    """More CR2 output strips than the device path takes (RSX_ERR_UNSUPPORTED): the
    patched Cr2Decompressor::decompress() runs its original body."""
    rng = np.random.default_rng(45)
    n_slices, sw = 70, 4
    W, H = n_slices * sw, 16
    src = C.smooth_image(rng, H, W)
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, [sw] * n_slices)
    blob, _, _, _ = synth.ljpeg_container(rows, 2, 14, [0, 0], [C.NIKON])
    (s0, a, e0), (s1, b, e1) = both(
        pair, lambda lib, img: lib.cr2_container(blob, img, n_slices, sw, sw), (W, H, 1),
        fwd=0, fell=1)
    assert s0 == 0 and s1 == 0, (e0, e1)
    assert np.array_equal(a, b) and np.array_equal(a[:, :W], src)
